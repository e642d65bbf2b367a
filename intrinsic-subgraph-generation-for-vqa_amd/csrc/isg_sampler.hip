// Node gate scores and the discrete top-k node-mask samplers (Gumbel relaxed top-k, I-MLE / AIMLE
// threshold top-k).  One wave owns one graph row of Nmax slots (real nodes first, then the 0.0
// pads of to_dense_batch, which compete exactly as in the reference); a lane holds SLOTS slots
// (slot j = s*64 + lane), so a row never leaves registers.
//
// Bit-exactness of the selected indices: log/exp inside the relaxed top-k are evaluated in fp64
// and rounded once to fp32 (i.e. correctly rounded fp32 results), divisions are IEEE fp32, and the
// softmax denominator is summed in fp64 -- the selection can then differ from a CPU run only where
// two khot values agree to within an ulp.  Ties are broken towards the lower slot index.
#include "isg_common.hpp"

namespace isg {

// ---- node gate --------------------------------------------------------------------------------------
// 16 lanes per node (4 nodes per wave); lane l covers float4 columns l, l+16, ...
__global__ __launch_bounds__(256) void node_gate_kernel(const float4 *__restrict__ xn, const float4 *__restrict__ q,
                                                        const int64_t *__restrict__ batch, int dbl,
                                                        float *__restrict__ gate, int N, int Q, float denom) {
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, l = lane & 15;
  const int n = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + grp;
  float part = 0.f;
  if (n < N) {
    int64_t b = batch[n];
    if (dbl) b = batch[min(b, (int64_t)N - 1)];   // batch[batch[n]] (quirk Q3)
    const float4 *xr = xn + (size_t)n * Q;
    const float4 *qr = q + (size_t)b * Q;
    for (int c = l; c < Q; c += 16) part += dot4_rn(xr[c], qr[c]);
  }
  const float dot = group_sum<16>(part);
  if (n < N && l == 0) gate[n] = gelu_libm(dot / denom);
}

// ---- row access shared by both samplers -----------------------------------------------------------------
struct RowArgs {
  const float *scores;
  const int *ptr;       // NULL -> dense [B, Nmax]
  const int *nmax_dev;  // NULL -> nmax_host
  const float *noise;   // NULL -> Philox
  float *out;
  float *khot_out;
  int B, nmax_host, k;
  float tau, noise_scale;
  uint64_t seed;
  const int *gids;      // NULL -> row b draws the Philox stream of graph b; else of graph gids[b] (a sub-batch keeps its graphs' streams)
};

template <int SLOTS>
__global__ __launch_bounds__(256) void topk_gumbel_kernel(RowArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;  // whole wave leaves together
  const int nmax = min(a.nmax_dev ? a.nmax_dev[0] : a.nmax_host, SLOTS * 64);
  const int base = a.ptr ? a.ptr[b] : b * nmax;
  const int n = a.ptr ? a.ptr[b + 1] - base : nmax;

  float flat[SLOTS], khot[SLOTS], onehot[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;
    float v = -INFINITY;
    if (j < nmax) {
      const float sc = j < n ? a.scores[base + j] : 0.f;                       // pad = 0.0 (quirk Q1)
      const float g = a.noise ? a.noise[(size_t)b * a.nmax_host + j]
                              : gumbel_from_bits(Philox::draw(a.seed, (uint32_t)(a.gids ? a.gids[b] : b), (uint32_t)j), 0.f, 1.f);
      v = sc + g;                                                              // gumbel_scheme.py:70
    }
    flat[s] = v;
    khot[s] = 0.f;
    onehot[s] = 0.f;
  }
  const int local_k = min(a.k, nmax);                                          // gumbel_scheme.py:58
  for (int it = 0; it < local_k; ++it) {                                       // :75-81
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const float km = fmaxf(1.0f - onehot[s], FLT_MIN);
      flat[s] = flat[s] + (float)log((double)km);
      mx = fmaxf(mx, flat[s] / a.tau);
    }
    mx = wave_max(mx);
    float ex[SLOTS];
    double sum = 0.0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      ex[s] = (float)exp((double)(flat[s] / a.tau - mx));
      sum += (double)ex[s];
    }
    const float den = (float)wave_sum_f64(sum);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      onehot[s] = ex[s] / den;
      khot[s] = khot[s] + onehot[s];
    }
  }
  // hard top-k of khot (ties -> lower slot index), straight-through value (hard - khot) + khot   :83-88
  float hard[SLOTS];
  bool taken[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) { hard[s] = 0.f; taken[s] = false; }
  for (int it = 0; it < local_k; ++it) {
    float best = -INFINITY;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
      if (!taken[s] && s * 64 + lane < nmax) best = fmaxf(best, khot[s]);
    const float wbest = wave_max(best);
    int idx = 0x7fffffff;
#pragma unroll
    for (int s = SLOTS - 1; s >= 0; --s)
      if (!taken[s] && s * 64 + lane < nmax && khot[s] == wbest) idx = s * 64 + lane;
    const int widx = wave_min_i(idx);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
      if (s * 64 + lane == widx) { taken[s] = true; hard[s] = 1.f; }
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;
    if (j < nmax) {
      if (a.khot_out) a.khot_out[(size_t)b * a.nmax_host + j] = khot[s];
      if (j < n) a.out[base + j] = (hard[s] - khot[s]) + khot[s];
    }
  }
}

template <int SLOTS>
__global__ __launch_bounds__(256) void topk_threshold_kernel(RowArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int nmax = min(a.nmax_dev ? a.nmax_dev[0] : a.nmax_host, SLOTS * 64);
  const int base = a.ptr ? a.ptr[b] : b * nmax;
  const int n = a.ptr ? a.ptr[b + 1] - base : nmax;

  float v[SLOTS];
  bool taken[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;
    float x = -INFINITY;
    if (j < nmax) {
      x = j < n ? a.scores[base + j] : 0.f;
      if (a.noise_scale != 0.f) {
        const float g = a.noise ? a.noise[(size_t)b * a.nmax_host + j]
                                : gumbel_from_bits(Philox::draw(a.seed, (uint32_t)(a.gids ? a.gids[b] : b), (uint32_t)j), 0.f, 0.3f);
        x = add_rn(x, mul_rn(g, a.noise_scale));                         // aimle.py:109,117 (mul, then add)
      }
    }
    v[s] = x;
    taken[s] = false;
  }
  float thresh = -INFINITY;
  const bool all = a.k >= nmax;                                                // deterministic_scheme.py:38-39
  if (!all) {
    for (int it = 0; it < a.k; ++it) {   // k-th largest, duplicates counted one at a time
      float best = -INFINITY;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (!taken[s] && s * 64 + lane < nmax) best = fmaxf(best, v[s]);
      thresh = wave_max(best);
      int idx = 0x7fffffff;
#pragma unroll
      for (int s = SLOTS - 1; s >= 0; --s)
        if (!taken[s] && s * 64 + lane < nmax && v[s] == thresh) idx = s * 64 + lane;
      const int widx = wave_min_i(idx);
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (s * 64 + lane == widx) taken[s] = true;
    }
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;
    const float z = (j < nmax && (all || v[s] >= thresh)) ? 1.f : 0.f;               // :41-42
    if (j < n && j < nmax) a.out[base + j] = z;
    if (a.khot_out && j < a.nmax_host) a.khot_out[(size_t)b * a.nmax_host + j] = z;  // dense row, pads included
  }
}

// Backward of the relaxed top-k (straight-through: d khot = d out, gumbel_scheme.py:83-90).  The forward recurrence
//   s_0 = scores + g;  p_i = softmax(s_i / tau);  s_{i+1} = s_i + log(max(1 - p_i, tiny));  khot = sum_i p_i
// is replayed with the forward kernel's arithmetic, p_i kept in LDS ([k][SLOTS*64] per wave), then walked backwards:
//   dp_i = G - ds_{i+1} / (1 - p_i)   (0 where the max() clamps),   ds_i = ds_{i+1} + p_i (dp_i - <dp_i, p_i>) / tau.
template <int SLOTS>
__global__ __launch_bounds__(256) void topk_gumbel_bwd_kernel(RowArgs a, const float *__restrict__ d_out,
                                                              float *__restrict__ d_scores) {
  extern __shared__ float s_hist[];
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int nmax = min(a.nmax_dev ? a.nmax_dev[0] : a.nmax_host, SLOTS * 64);
  const int base = a.ptr ? a.ptr[b] : b * nmax;
  const int n = a.ptr ? a.ptr[b + 1] - base : nmax;
  const int local_k = min(a.k, nmax);
  float *hist = s_hist + (size_t)(threadIdx.x >> 6) * a.k * SLOTS * 64;

  float flat[SLOTS], onehot[SLOTS], G[SLOTS], ds[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;
    float v = -INFINITY;
    if (j < nmax) {
      const float sc = j < n ? a.scores[base + j] : 0.f;
      const float g = a.noise ? a.noise[(size_t)b * a.nmax_host + j]
                              : gumbel_from_bits(Philox::draw(a.seed, (uint32_t)(a.gids ? a.gids[b] : b), (uint32_t)j), 0.f, 1.f);
      v = sc + g;
    }
    flat[s] = v;
    onehot[s] = 0.f;
    G[s] = (j < n && j < nmax) ? d_out[base + j] : 0.f;
    ds[s] = 0.f;
  }
  for (int it = 0; it < local_k; ++it) {
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const float km = fmaxf(1.0f - onehot[s], FLT_MIN);
      flat[s] = flat[s] + (float)log((double)km);
      mx = fmaxf(mx, flat[s] / a.tau);
    }
    mx = wave_max(mx);
    float ex[SLOTS];
    double sum = 0.0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      ex[s] = (float)exp((double)(flat[s] / a.tau - mx));
      sum += (double)ex[s];
    }
    const float den = (float)wave_sum_f64(sum);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      onehot[s] = ex[s] / den;
      hist[(it * SLOTS + s) * 64 + lane] = onehot[s];
    }
  }
  for (int it = local_k - 1; it >= 0; --it) {
    float p[SLOTS], dp[SLOTS];
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      p[s] = hist[(it * SLOTS + s) * 64 + lane];
      const float om = 1.0f - p[s];
      dp[s] = G[s] - (om > FLT_MIN ? ds[s] / om : 0.f);
      part += dp[s] * p[s];
    }
    const float dot = wave_sum(part);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) ds[s] += p[s] * (dp[s] - dot) / a.tau;
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int j = s * 64 + lane;
    if (j < n && j < nmax) d_scores[base + j] = ds[s];
  }
}

static int pick_slots(int nmax_host) {
  if (nmax_host <= 64) return 1;
  if (nmax_host <= 128) return 2;
  if (nmax_host <= 256) return 4;
  if (nmax_host <= 512) return 8;
  if (nmax_host <= 1024) return 16;
  return 0;
}

}  // namespace isg

using namespace isg;

extern "C" int isg_node_gate(const float *xn, const float *q, const int64_t *batch, int32_t double_index, float *gate,
                             int64_t N, int32_t C, void *stream) {
  if (N < 0 || C <= 0) return ISG_EINVAL;
  if (N == 0) return ISG_OK;
  if (!xn || !q || !batch || !gate) return ISG_EINVAL;
  if ((C & 3) != 0 || N >= (1ll << 31)) return ISG_EUNSUPPORTED;
  const float denom = sqrtf((float)C);   // torch.sqrt(torch.tensor(C)) -> fp32 (masking.py:153)
  node_gate_kernel<<<(unsigned)((N + 15) / 16), 256, 0, as_stream(stream)>>>(
      (const float4 *)xn, (const float4 *)q, batch, double_index, gate, (int)N, C >> 2, denom);
  return check_launch();
}

static int check_rows(const float *scores, int64_t B, int32_t nmax_host, int32_t k, float *out) {
  if (B < 0 || nmax_host < 0 || k < 0) return ISG_EINVAL;
  if (B > 0 && nmax_host > 0 && (!scores || !out)) return ISG_EINVAL;
  if (B >= (1ll << 31)) return ISG_EUNSUPPORTED;
  return ISG_OK;
}

extern "C" int isg_topk_gumbel(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax_host,
                               const int32_t *nmax_dev, const float *noise, uint64_t seed, const int32_t *graph_ids,
                               int32_t k, float tau, float *out, float *khot_out, void *stream) {
  int st = check_rows(scores, B, nmax_host, k, out);
  if (st != ISG_OK) return st;
  if (B == 0 || nmax_host == 0) return ISG_OK;
  if (!(tau > 0.f)) return ISG_EINVAL;
  RowArgs a{scores, ptr, nmax_dev, noise, out, khot_out, (int)B, nmax_host, k, tau, 0.f, seed, graph_ids};
  dim3 grid((unsigned)((B + 3) / 4)), block(256);
  hipStream_t s = as_stream(stream);
  switch (pick_slots(nmax_host)) {
    case 1: topk_gumbel_kernel<1><<<grid, block, 0, s>>>(a); break;
    case 2: topk_gumbel_kernel<2><<<grid, block, 0, s>>>(a); break;
    case 4: topk_gumbel_kernel<4><<<grid, block, 0, s>>>(a); break;
    case 8: topk_gumbel_kernel<8><<<grid, block, 0, s>>>(a); break;
    case 16: topk_gumbel_kernel<16><<<grid, block, 0, s>>>(a); break;
    default: return ISG_EUNSUPPORTED;
  }
  return check_launch();
}

extern "C" int isg_topk_threshold(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax_host,
                                  const int32_t *nmax_dev, const float *noise, float noise_scale, uint64_t seed,
                                  const int32_t *graph_ids, int32_t k, float *out, float *dense_out, void *stream) {
  int st = check_rows(scores, B, nmax_host, k, out);
  if (st != ISG_OK) return st;
  if (B == 0 || nmax_host == 0) return ISG_OK;
  RowArgs a{scores, ptr, nmax_dev, noise, out, dense_out, (int)B, nmax_host, k, 1.f, noise_scale, seed, graph_ids};
  dim3 grid((unsigned)((B + 3) / 4)), block(256);
  hipStream_t s = as_stream(stream);
  switch (pick_slots(nmax_host)) {
    case 1: topk_threshold_kernel<1><<<grid, block, 0, s>>>(a); break;
    case 2: topk_threshold_kernel<2><<<grid, block, 0, s>>>(a); break;
    case 4: topk_threshold_kernel<4><<<grid, block, 0, s>>>(a); break;
    case 8: topk_threshold_kernel<8><<<grid, block, 0, s>>>(a); break;
    case 16: topk_threshold_kernel<16><<<grid, block, 0, s>>>(a); break;
    default: return ISG_EUNSUPPORTED;
  }
  return check_launch();
}

extern "C" int isg_topk_gumbel_bwd(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax_host,
                                   const int32_t *nmax_dev, const float *noise, uint64_t seed, const int32_t *graph_ids,
                                   int32_t k, float tau, const float *d_out, float *d_scores, void *stream) {
  int st = check_rows(scores, B, nmax_host, k, d_scores);
  if (st != ISG_OK) return st;
  if (B == 0 || nmax_host == 0) return ISG_OK;
  if (!(tau > 0.f) || !d_out) return ISG_EINVAL;
  const int slots = pick_slots(nmax_host);
  if (slots == 0) return ISG_EUNSUPPORTED;
  const size_t lds = (size_t)4 * k * slots * 64 * sizeof(float);
  if (lds > 64 * 1024) return ISG_EUNSUPPORTED;   // k * row length beyond the LDS history
  RowArgs a{scores, ptr, nmax_dev, noise, nullptr, nullptr, (int)B, nmax_host, k, tau, 0.f, seed, graph_ids};
  dim3 grid((unsigned)((B + 3) / 4)), block(256);
  hipStream_t s = as_stream(stream);
  switch (slots) {
    case 1: topk_gumbel_bwd_kernel<1><<<grid, block, lds, s>>>(a, d_out, d_scores); break;
    case 2: topk_gumbel_bwd_kernel<2><<<grid, block, lds, s>>>(a, d_out, d_scores); break;
    case 4: topk_gumbel_bwd_kernel<4><<<grid, block, lds, s>>>(a, d_out, d_scores); break;
    case 8: topk_gumbel_bwd_kernel<8><<<grid, block, lds, s>>>(a, d_out, d_scores); break;
    case 16: topk_gumbel_bwd_kernel<16><<<grid, block, lds, s>>>(a, d_out, d_scores); break;
  }
  return check_launch();
}
