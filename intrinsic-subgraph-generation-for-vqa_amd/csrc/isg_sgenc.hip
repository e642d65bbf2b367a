// Scene-graph encoder without the concatenations (reference: ISubGVQA/models/scene_graph_encoder.py:108-143).
//
// The reference's EdgeModel / NodeModel apply a Linear to cat([x[row], x[col], e]) (E x 900) and to cat([x[row], e'])
// (E x 600).  A Linear over a concatenation is a sum of Linears over its parts,
//     W [x_src; x_dst; e] = W_a x_src + W_b x_dst + W_c e,
// and the node parts depend on the NODE only: they are projected once per node (N rows instead of E), the edge-token
// part W_c emb[token] is a lookup in a [vocabulary, C] table projected once per forward (the `added_sym_edge` sign flip
// commutes with a linear map), and what is left per edge is this kernel: gather the rows, add, bias, exact GELU.
// The [E, 900] / [E, 600] tensors never exist and the flops drop 1.67x.
//
//   out[e] = act( A[ia[e]] + B[ib[e]] + sign[e] * T[it[e]] + D[e] + bias )      (every operand but A optional)
//
// one thread = one float4 of one output row; consecutive threads walk a row, so every row access is coalesced; the
// gathered rows of a scene graph are neighbours in memory (PyG batching) and are served by L2.
#include "isg_f16x3.hpp"

namespace isg {

struct GatherAddArgs {
  const float4 *A;
  const int64_t *ia;
  const float4 *B;
  const int64_t *ib;
  const float4 *T;
  const int64_t *it;
  const float *sign;
  const float4 *D;
  const float4 *bias;
  float4 *out;
  _Float16 *planes;       // the rows as the planes32 operand of isg_linear_h3p (csrc/isg_gemm_h3p.hip), or NULL
  float *planes_inv;
  int64_t E;
  int Q;        // float4 per row
  int lda, ldb, ldt, ldd;   // row strides in float4
  int act;
};

__device__ __forceinline__ float4 gather_add_value(const GatherAddArgs &a, int64_t e, int c) {
  float4 v = a.A[(size_t)a.ia[e] * a.lda + c];
  if (a.B) {
    const float4 b = a.B[(size_t)a.ib[e] * a.ldb + c];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (a.T) {
    const float4 w = a.T[(size_t)a.it[e] * a.ldt + c];
    const float s = a.sign ? a.sign[e] : 1.f;
    v.x += s * w.x; v.y += s * w.y; v.z += s * w.z; v.w += s * w.w;
  }
  if (a.D) {
    const float4 d = a.D[(size_t)e * a.ldd + c];
    v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
  }
  if (a.bias) {
    const float4 b = a.bias[c];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (a.act == 1) { v.x = gelu_exact(v.x); v.y = gelu_exact(v.y); v.z = gelu_exact(v.z); v.w = gelu_exact(v.w); }
  return v;
}

__global__ __launch_bounds__(256) void gather_add_kernel(GatherAddArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.E * a.Q) return;
  const int64_t e = t / a.Q;
  const int c = (int)(t - e * a.Q);
  a.out[(size_t)e * a.Q + c] = gather_add_value(a, e, c);
}

// The same rows written as planes32 (+ fp32 rows where `out` is given): the Linear that reads them (edge_mlp.2 / node_mlp_1.2,
// K = 300 over E rows) needs no isg_split_planes32 pass -- at 205 k edges that pass was 126 us behind a 141 us kernel.  Sixteen
// lanes per row (C = 300 is 75 float4: five passes of 16 lanes keep 94 % of the lane slots busy, a whole wave per row 59 %), the
// row held in registers between its largest magnitude and its split (PMAX passes: C <= 512; wider rows are evaluated twice).
template <int PMAX>
__global__ __launch_bounds__(256) void gather_add_planes32_kernel(GatherAddArgs a) {
  const int64_t e = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  if (e >= a.E) return;
  const int KT = (a.Q + 7) >> 3;
  float4 v[PMAX > 0 ? PMAX : 1];
  float mx = 0.f;
  if constexpr (PMAX > 0) {
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
      const int c = l + 16 * p;
      v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < a.Q) v[p] = gather_add_value(a, e, c);
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[p].x), fabsf(v[p].y)), fmaxf(fabsf(v[p].z), fabsf(v[p].w))));
    }
  } else {
    for (int c = l; c < a.Q; c += 16) {
      const float4 t = gather_add_value(a, e, c);
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))));
    }
  }
  mx = group_max<16>(mx);
  float s, inv;
  h3_scale(mx, s, inv);
  if (l == 0) a.planes_inv[e] = inv;
  _Float16 *pl = a.planes + e * KT * 64;
  auto put = [&](int c, float4 t) {
    if (a.out && c < a.Q) a.out[(size_t)e * a.Q + c] = t;
    t.x *= s; t.y *= s; t.z *= s; t.w *= s;
    const hf16x4 hi = {(_Float16)t.x, (_Float16)t.y, (_Float16)t.z, (_Float16)t.w};
    const hf16x4 mid = {(_Float16)(t.x - (float)hi[0]), (_Float16)(t.y - (float)hi[1]), (_Float16)(t.z - (float)hi[2]),
                        (_Float16)(t.w - (float)hi[3])};
    _Float16 *d = pl + (c >> 3) * 64 + (c & 7) * 4;
    *reinterpret_cast<hf16x4 *>(d) = hi;
    *reinterpret_cast<hf16x4 *>(d + 32) = mid;
  };
  if constexpr (PMAX > 0) {
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
      if (l + 16 * p < KT * 8) put(l + 16 * p, v[p]);           // beyond Q: the zeros of the k padding
  } else {
    for (int c = l; c < KT * 8; c += 16) put(c, c < a.Q ? gather_add_value(a, e, c) : make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

}  // namespace isg

using namespace isg;

extern "C" int isg_gather_add(const float *A, const int64_t *ia, int32_t lda, const float *B, const int64_t *ib, int32_t ldb,
                              const float *T, const int64_t *it, const float *sign, int32_t ldt, const float *D,
                              int32_t ldd, const float *bias, float *out, int64_t E, int32_t C, int32_t act,
                              uint16_t *planes, float *planes_inv, void *stream) {
  if (E < 0 || C <= 0 || act < 0 || act > 1) return ISG_EINVAL;
  if (E == 0) return ISG_OK;
  if (!A || !ia || (!out && !planes) || (B && !ib) || (T && !it) || (!planes != !planes_inv)) return ISG_EINVAL;
  if ((C & 3) || (lda & 3) || (B && (ldb & 3)) || (T && (ldt & 3)) || (D && (ldd & 3))) return ISG_EUNSUPPORTED;
  const int64_t total = E * (C >> 2);
  if ((total + 255) / 256 >= (1ll << 31)) return ISG_EUNSUPPORTED;
  GatherAddArgs a{(const float4 *)A, ia, (const float4 *)B, ib, (const float4 *)T, it, sign, (const float4 *)D,
                  (const float4 *)bias, (float4 *)out, reinterpret_cast<_Float16 *>(planes), planes_inv, E, C >> 2, lda >> 2,
                  ldb >> 2, ldt >> 2, ldd >> 2, act};
  if (planes) {
    if ((reinterpret_cast<uintptr_t>(planes) & 15) != 0 || (E + 15) / 16 >= (1ll << 31)) return ISG_EUNSUPPORTED;
    const unsigned grid = (unsigned)((E + 15) / 16);
    const int passes = (((C >> 2) + 7) / 8 * 8 + 15) / 16;       // float4 of a padded row over 16 lanes
    if (passes <= 5) gather_add_planes32_kernel<5><<<grid, 256, 0, as_stream(stream)>>>(a);
    else if (passes <= 8) gather_add_planes32_kernel<8><<<grid, 256, 0, as_stream(stream)>>>(a);
    else gather_add_planes32_kernel<0><<<grid, 256, 0, as_stream(stream)>>>(a);
  } else {
    gather_add_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(a);
  }
  return check_launch();
}
