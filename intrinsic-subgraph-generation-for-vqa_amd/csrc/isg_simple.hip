// SIMPLE sampler: exact k-subset marginals through the reference's exactly-k circuit + a Gumbel top-k sample.
//
// Reference behaviour: EdgeSIMPLEBatched.forward (policy 'edge_candid'), ISubGVQA/sampling/methods/simple_scheme.py:44-162;
// Layer.log_pr / sample, simple.py:203-251; circuit construction, create_simple_constraint.py:34-73.
// The circuit is a balanced binary tree over n = 2^ceil(log2 Nmax) variables whose node (level l, block i, count j)
// means "exactly j of the block's 2^l variables are on".  All blocks of a level are alike, so the reference's per-node
// index tensors collapse into (level, count) tables built on the host (SimpleTables): reachability from the root,
// number of elements, number of parents.  Two padding accidents of the reference are part of the function (they decide
// the result whenever a row holds more zero-score pads than k, i.e. on most ragged batches) and are reproduced:
// element lists are padded with a dummy of log-weight -1000 (an impossible node evaluates to ~ -2000, not -inf) and
// parent lists are padded with the same dummy.
//
// One wave owns one row; the tree lives in LDS: W (log-weights, bottom-up) and M (log-marginals, top-down), each
// (2n - 1)(k + 1) floats; theta (normalised element weights) is recomputed from W on the way down instead of stored.
// logsumexp follows torch (an infinite maximum is replaced by 0 before the subtraction), NaNs propagate as in IEEE.
#include "isg_common.hpp"

namespace isg {

constexpr int SM_MAXL = 10;   // n <= 1024
constexpr int SM_MAXK = 16;

struct SimpleTables {
  int n, k, levels, max_elements, max_parents;
  int cap[SM_MAXL + 1];
  unsigned reach[SM_MAXL + 1];                 // bit j: node (l, *, j) reachable
  unsigned char n_par[SM_MAXL + 1][SM_MAXK + 1];
};

struct SimpleArgs {
  const float *scores;
  const int *ptr;        // NULL -> dense [B, nmax]
  const float *uniform;  // [B, n] torch.rand draw, or NULL -> Philox
  float *out;            // mask in the layout of scores
  float *marg_out;       // optional dense [B, nmax]
  int B, nmax;
  uint64_t seed;
  const int *gids;       // NULL -> row b draws graph b's stream; else graph gids[b]'s
};

__device__ __forceinline__ float sm_log1mexp(float x) {   // simple.py:45-57: log(1 - exp(-|x|))
  x = -fabsf(x);
  return x > -0.6931471805599453094f ? logf(-expm1f(x)) : log1pf(-expf(x));
}

// torch.logsumexp over `cnt` values produced by f(t), plus `pads` copies of `padv`
template <typename F>
__device__ __forceinline__ float sm_lse(int cnt, F f, int pads, float padv) {
  float m = pads > 0 ? padv : -INFINITY;
  for (int t = 0; t < cnt; ++t) {
    const float v = f(t);
    m = (v > m || v != v) ? v : m;            // amax propagates NaN
  }
  const float ms = (fabsf(m) == INFINITY) ? 0.f : m;
  float s = pads > 0 ? (float)pads * expf(padv - ms) : 0.f;
  for (int t = 0; t < cnt; ++t) s += expf(f(t) - ms);
  return logf(s) + ms;
}

__global__ __launch_bounds__(256) void simple_kernel(SimpleArgs a, SimpleTables c, int rows_per_block) {
  extern __shared__ float s_tree[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= rows_per_block) return;
  const int b = blockIdx.x * rows_per_block + wave;
  if (b >= a.B) return;
  const int n = c.n, k = c.k, K1 = k + 1, L = c.levels;
  const int tree = (2 * n - 1) * K1;
  float *W = s_tree + (size_t)wave * 2 * tree, *M = W + tree;
  const int base = a.ptr ? a.ptr[b] : b * a.nmax;
  const int len = a.ptr ? a.ptr[b + 1] - base : a.nmax;
  auto lvl = [&](int l) { return (2 * n - (2 * n >> l)) * K1; };   // offset of level l (n >> l blocks)
  auto flat = [&](int i) -> float {
    return i < len ? a.scores[base + i] : (i < a.nmax ? 0.f : -1.0e10f);   // to_dense_batch pad 0.0, then -LARGE_NUMBER
  };
  // leaves
  for (int i = lane; i < n; i += 64) {
    const float w = flat(i);
    W[i * K1 + 0] = sm_log1mexp(-w);
    W[i * K1 + 1] = w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // bottom-up: W(l, i, j) = logsumexp_jj (W(l-1, 2i, jj) + W(l-1, 2i+1, j-jj)), dummy-padded to max_elements
  for (int l = 1; l <= L; ++l) {
    const int blocks = n >> l, capl = c.cap[l], capc = c.cap[l - 1];
    const float *Wc = W + lvl(l - 1);
    float *Wl = W + lvl(l);
    for (int idx = lane; idx < blocks * (capl + 1); idx += 64) {
      const int i = idx / (capl + 1), j = idx - i * (capl + 1);
      if (!((c.reach[l] >> j) & 1)) continue;
      const int lo = max(0, j - capc), hi = min(j, capc);
      const float *pl = Wc + (2 * i) * K1, *pr = Wc + (2 * i + 1) * K1;
      Wl[i * K1 + j] = sm_lse(hi - lo + 1, [&](int t) { return pl[lo + t] + pr[j - lo - t]; },
                              c.max_elements - (hi - lo + 1), -2000.f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // top-down: M(root) = 0; M(child) = logsumexp over parents (theta(parent, element) + M(parent)), dummy-padded
  if (lane == 0) M[lvl(L) + k] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int l = L - 1; l >= 0; --l) {
    const int blocks = n >> l, capl = c.cap[l], capp = c.cap[l + 1];
    const float *Wc = W + lvl(l), *Wp = W + lvl(l + 1), *Mp = M + lvl(l + 1);
    float *Ml = M + lvl(l);
    for (int idx = lane; idx < blocks * (capl + 1); idx += 64) {
      const int i = idx / (capl + 1), j = idx - i * (capl + 1);
      if (!((c.reach[l] >> j) & 1)) continue;
      const int pi = i >> 1;
      const float *pl = Wc + (2 * pi) * K1, *pr = Wc + (2 * pi + 1) * K1;
      // parents jp in [j, min(capp, j + capl)] that are reachable; gather them into a tiny list first
      int jps[SM_MAXK + 1], np = 0;
      for (int jp = j; jp <= min(capp, j + capl); ++jp)
        if ((c.reach[l + 1] >> jp) & 1) jps[np++] = jp;
      Ml[i * K1 + j] = sm_lse(np, [&](int t) {
        const int jp = jps[t];
        const int jj = (i & 1) ? jp - j : j;          // this child is the sub (right) / the prime (left)
        const float theta = (pl[jj] + pr[jp - jj]) - Wp[pi * K1 + jp];
        return theta + Mp[pi * K1 + jp];
      }, c.max_parents - np, -1000.f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // Gumbel top-k sample of the raw scores (simple.py:99-118): keys = w + (-log(-log(u))), k largest -> 1
  // keys live in W's leaf slots 0 (no longer needed), marginals are exp(M(0, i, 1))
  for (int i = lane; i < n; i += 64) {
    const float u = a.uniform ? a.uniform[(size_t)b * n + i] : (float)(Philox::draw(a.seed, (uint32_t)(a.gids ? a.gids[b] : b), (uint32_t)i) >> 8) * (1.0f / 16777216.0f);
    W[i * K1 + 0] = flat(i) + (-logf(-logf(u)));
    W[i * K1 + 1] = 0.f;                                // hot flag
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int it = 0; it < k; ++it) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
      const float v = W[i * K1 + 0];
      if (W[i * K1 + 1] == 0.f && (v > best || (v == best && i < bi))) { best = v; bi = i; }
    }
    const float wb = wave_max(best);
    const int wi = wave_min_i(best == wb ? bi : 0x7fffffff);
    if (lane == 0 && wi < n) W[wi * K1 + 1] = 1.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i < a.nmax; i += 64) {
    const float marg = expf(M[i * K1 + 1]);
    if (a.marg_out) a.marg_out[(size_t)b * a.nmax + i] = marg;
    if (i < len) a.out[base + i] = (W[i * K1 + 1] - marg) + marg;          // simple_scheme.py:130
  }
}

}  // namespace isg

using namespace isg;

extern "C" int isg_simple_topk(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax, const float *uniform,
                               uint64_t seed, const int32_t *graph_ids, int32_t k, float *out, float *marg_out, void *stream) {
  if (B < 0 || nmax < 0 || k <= 0) return ISG_EINVAL;
  if (B == 0 || nmax == 0) return ISG_OK;
  if (!scores || !out) return ISG_EINVAL;
  if (B >= (1ll << 31) || nmax > (1 << SM_MAXL)) return ISG_EUNSUPPORTED;
  SimpleTables c{};
  int L = 0;
  while ((1 << L) < nmax) ++L;
  c.n = 1 << L;
  c.k = k < nmax ? k : nmax;                        // simple_scheme.py:84
  c.levels = L;
  if (c.k > SM_MAXK) return ISG_EUNSUPPORTED;
  // the (level, count) tables of the exactly-k circuit (create_simple_constraint.py:34-66, simple.py:132-200)
  int n_elem[SM_MAXL + 1][SM_MAXK + 1] = {};
  for (int l = 0; l <= L; ++l) c.cap[l] = c.k < (1 << l) ? c.k : (1 << l);
  for (int l = 1; l <= L; ++l)
    for (int j = 0; j <= c.cap[l]; ++j)
      for (int jj = 0; jj <= j; ++jj)
        if (jj <= c.cap[l - 1] && j - jj <= c.cap[l - 1]) ++n_elem[l][j];
  c.reach[L] = 1u << c.k;
  for (int l = L - 1; l >= 0; --l)
    for (int j = 0; j <= c.cap[l]; ++j)
      for (int jp = j; jp <= c.cap[l + 1] && jp - j <= c.cap[l]; ++jp)
        if ((c.reach[l + 1] >> jp) & 1) { c.reach[l] |= 1u << j; ++c.n_par[l][j]; }
  for (int l = 1; l <= L; ++l)
    for (int j = 0; j <= c.cap[l]; ++j)
      if (((c.reach[l] >> j) & 1) && n_elem[l][j] > c.max_elements) c.max_elements = n_elem[l][j];
  for (int l = 0; l < L; ++l)
    for (int j = 0; j <= c.cap[l]; ++j)
      if (((c.reach[l] >> j) & 1) && c.n_par[l][j] > c.max_parents) c.max_parents = c.n_par[l][j];
  const size_t row_bytes = (size_t)2 * (2 * c.n - 1) * (c.k + 1) * sizeof(float);
  int rows = (int)((48 * 1024) / row_bytes);
  if (rows > 4) rows = 4;
  if (rows < 1) rows = 1;
  if (row_bytes > 150 * 1024) return ISG_EUNSUPPORTED;
  SimpleArgs a{scores, ptr, uniform, out, marg_out, (int)B, nmax, seed, graph_ids};
  if (rows * row_bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(simple_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(rows * row_bytes)) != hipSuccess)
    return ISG_ELAUNCH;
  simple_kernel<<<(unsigned)((B + rows - 1) / rows), 256, rows * row_bytes, as_stream(stream)>>>(a, c, rows);
  return check_launch();
}
