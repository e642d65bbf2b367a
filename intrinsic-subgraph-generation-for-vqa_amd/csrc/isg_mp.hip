// GATv2 message passing with edge features over a CSR-by-destination plan, plus the small
// gather/scatter companions of the layer (instruction gate, node->edge mask, scatter-mean).
//
// Work decomposition of the message-passing kernel (gfx950, wave = 64):
//   * one workgroup (4 waves) owns MP_NPB consecutive destination nodes; their CSR segment
//     (row pointers, source ids, original edge ids) is staged once into LDS with coalesced loads,
//     so the per-edge loop has no dependent index loads from HBM
//   * one wave owns one destination node at a time; the 64 lanes are split into H groups of
//     G = 64/H lanes, group g = head g, lane l of a group owns float4 columns l, l+G, ... of that
//     head -> every row access is H contiguous 16*G-byte pieces (256 B for H = 4)
//   * per-head attention logits are a G-lane DPP butterfly (no LDS), kept in an LDS strip per
//     wave (overflow for in-degree > MP_LCAP goes through the alpha output buffer)
//   * softmax is the exact three-step form of torch_geometric.utils.softmax (max, exp-sum + 1e-16,
//     divide) and the aggregation adds messages in ascending edge id, i.e. in the summation order of
//     the reference's CPU scatter, ONE fma per term (a single rounding where the CPU's multiply and
//     add round twice; written as fmaf so that it does not depend on the compiler's contraction)
#include "isg_mp.hpp"

#include <stdlib.h>

#define ISG_MP_DEFAULT_FLAGS (1 | 4)   // measured inside the layer pipeline (profiles/r01_c, r01_r), not on cold caches

namespace isg {

template <int H, int P>
__global__ __launch_bounds__(MP_WAVES * 64) void gatv2_mp_kernel(MpArgs a) {
  constexpr int G = 64 / H;
  __shared__ int s_rowptr[MP_NPB + 1];
  __shared__ int s_src[MP_ECAP];
  __shared__ int s_eid[MP_ECAP];
  __shared__ float s_logit[MP_WAVES][MP_LCAP * H];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware chunk order: workgroups b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous range of node chunks -- the two chunks a graph straddles then hit the same L2
  int chunk = blockIdx.x;
  if (a.flags & 2) {
    const int cpx = gridDim.x >> 3;
    chunk = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    if (chunk >= a.nchunks) return;
  }
  const bool nt = a.flags & 1;
  const int n0 = chunk * MP_NPB;
  const int nn = min(MP_NPB, a.N - n0);
  if (tid <= nn) s_rowptr[tid] = a.rowptr[n0 + tid];
  __syncthreads();
  const int e0 = s_rowptr[0];
  const int ne = min(s_rowptr[nn] - e0, MP_ECAP);
  for (int t = tid; t < ne; t += MP_WAVES * 64) {
    s_src[t] = a.src[e0 + t];
    s_eid[t] = a.eid[e0 + t];
  }
  __syncthreads();

  const int g = lane / G, l = lane % G;
  const int Q = a.C >> 2;   // float4 per head
  const int R = H * Q;      // float4 per row
  int off[P];
  bool ok[P];
  float4 att4[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int q = p * G + l;
    ok[p] = q < Q;
    off[p] = g * Q + (ok[p] ? q : 0);
    att4[p] = ok[p] ? a.att[off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int mode = a.edge_mask ? 2 : (a.node_mask ? 1 : 0);
  float *s_lg = s_logit[wave];

  for (int k = wave; k < nn; k += MP_WAVES) {
    const int i = n0 + k;
    const int rb = s_rowptr[k], re = s_rowptr[k + 1];
    float4 xr4[P];
#pragma unroll
    for (int p = 0; p < P; ++p) xr4[p] = ok[p] ? a.x_r[(size_t)i * a.ldr4 + off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float mi = mode == 1 ? a.node_mask[i] : 1.f;

    // ---- pass 1: logits a[e,h] and their per-head maximum --------------------------------------
    float mx = -INFINITY;
    for (int t = rb; t < re; ++t) {
      const int rel = t - e0;
      int j, e;
      if (rel < MP_ECAP) { j = s_src[rel]; e = s_eid[rel]; } else { j = a.src[t]; e = a.eid[t]; }
      float me = 1.f;
      if (mode == 1) me = a.node_mask[j] * mi;
      else if (mode == 2) me = a.edge_mask[e];
      const float4 *xl = a.x_l + (size_t)j * a.ldl4;
      const float4 *ep = a.e_proj + (size_t)e * a.lde4;
      float part = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          float4 u = xl[off[p]], v = ld_stream(ep + off[p], nt), s;
          s.x = (xr4[p].x + u.x) + v.x;
          s.y = (xr4[p].y + u.y) + v.y;
          s.z = (xr4[p].z + u.z) + v.z;
          s.w = (xr4[p].w + u.w) + v.w;
          if (mode != 0) { s.x *= me; s.y *= me; s.z *= me; s.w *= me; }
          s.x = leaky(s.x, a.slope); s.y = leaky(s.y, a.slope); s.z = leaky(s.z, a.slope); s.w = leaky(s.w, a.slope);
          if (mode != 0) { s.x *= me; s.y *= me; s.z *= me; s.w *= me; }
          part += dot4(s, att4[p]);
        }
      }
      const float logit = group_sum<G>(part);
      mx = fmaxf(mx, logit);
      const int slot = t - rb;
      if (l == 0) {
        if (slot < MP_LCAP) s_lg[slot * H + g] = logit;
        else a.alpha[(size_t)e * H + g] = logit;
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- pass 2: denominator, accumulated in edge order like the CPU scatter_sum ---------------
    float den = 0.f;
    for (int t = rb; t < re; ++t) {
      const int slot = t - rb;
      float lg;
      if (slot < MP_LCAP) {
        lg = s_lg[slot * H + g];
      } else {
        const int rel = t - e0;
        const int e = rel < MP_ECAP ? s_eid[rel] : a.eid[t];
        lg = l == 0 ? a.alpha[(size_t)e * H + g] : 0.f;   // read back by the lane that wrote it
        lg = __shfl(lg, lane - l, 64);
      }
      den += expf(lg - mx);
    }
    den += 1e-16f;

    // ---- pass 3: alpha out + weighted aggregation (edge-id order, one fma per term) ------
    float4 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = rb; t < re; ++t) {
      const int rel = t - e0;
      int j, e;
      if (rel < MP_ECAP) { j = s_src[rel]; e = s_eid[rel]; } else { j = a.src[t]; e = a.eid[t]; }
      const int slot = t - rb;
      float lg;
      if (slot < MP_LCAP) {
        lg = s_lg[slot * H + g];
      } else {
        lg = l == 0 ? a.alpha[(size_t)e * H + g] : 0.f;
        lg = __shfl(lg, lane - l, 64);
      }
      const float w = expf(lg - mx) / den;
      if (l == 0) a.alpha[(size_t)e * H + g] = w;
      float wm = w;
      if (mode == 1) wm = mul_rn(w, a.node_mask[j] * mi);
      else if (mode == 2) wm = mul_rn(w, a.edge_mask[e]);
      const float4 *xl = a.x_l + (size_t)j * a.ldl4;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          float4 u = xl[off[p]];
          acc[p].x = fmaf(u.x, wm, acc[p].x);
          acc[p].y = fmaf(u.y, wm, acc[p].y);
          acc[p].z = fmaf(u.z, wm, acc[p].z);
          acc[p].w = fmaf(u.w, wm, acc[p].w);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (ok[p]) {
        float4 o = acc[p];
        if (a.bias) {
          float4 b = a.bias[off[p]];
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        st_stream(a.out + (size_t)i * R + off[p], o, nt);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int H>
static int launch_mp(const MpArgs &a, hipStream_t st) {
  constexpr int G = 64 / H;
  const int Q = a.C >> 2;
  const int P = (Q + G - 1) / G;
  const int blocks = (a.N + MP_NPB - 1) / MP_NPB;
  MpArgs b = a;
  b.nchunks = blocks;
  const int gridx = (b.flags & 2) ? ((blocks + 7) / 8) * 8 : blocks;
  dim3 grid(gridx), block(MP_WAVES * 64);
  switch (P) {
    case 1: gatv2_mp_kernel<H, 1><<<grid, block, 0, st>>>(b); break;
    case 2: gatv2_mp_kernel<H, 2><<<grid, block, 0, st>>>(b); break;
    case 3: gatv2_mp_kernel<H, 3><<<grid, block, 0, st>>>(b); break;
    case 4: gatv2_mp_kernel<H, 4><<<grid, block, 0, st>>>(b); break;
    case 5: gatv2_mp_kernel<H, 5><<<grid, block, 0, st>>>(b); break;
    case 6: gatv2_mp_kernel<H, 6><<<grid, block, 0, st>>>(b); break;
    case 7: gatv2_mp_kernel<H, 7><<<grid, block, 0, st>>>(b); break;
    case 8: gatv2_mp_kernel<H, 8><<<grid, block, 0, st>>>(b); break;
    default: return ISG_EUNSUPPORTED;
  }
  return check_launch();
}

// ---- instruction gate: gelu(x * instr[batch]) ---------------------------------------------------
__global__ void instr_gate_kernel(const float4 *__restrict__ x, const float4 *__restrict__ instr,
                                  const int64_t *__restrict__ batch, float4 *__restrict__ out, int64_t total,
                                  int Q) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int64_t n = idx / Q;
  int q = (int)(idx - n * Q);
  int64_t b = batch[n];
  float4 v = x[idx], w = instr[b * Q + q];
  const isg_f32x2 g0 = gelu_exact2(isg_f32x2{v.x * w.x, v.y * w.y}), g1 = gelu_exact2(isg_f32x2{v.z * w.z, v.w * w.w});
  out[idx] = make_float4(g0.x, g0.y, g1.x, g1.y);
}

// cat((a, b, a * b), dim=1) with the row maxima of the result beside it (isubgvqa.py:288-291: the classifier head's input).
// 32 lanes per row, float4 pieces; the maxima feed the row scales of the Linear that reads the result (isg_linear_f16x3_tile),
// which would otherwise make its own pass over it.
__global__ __launch_bounds__(256) void cat_mul_rowmax_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             float *__restrict__ out, float *__restrict__ rowmax, int M, int Q) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (row >= M) return;
  const float4 *ar = reinterpret_cast<const float4 *>(a) + (size_t)row * Q;
  const float4 *br = reinterpret_cast<const float4 *>(b) + (size_t)row * Q;
  float4 *o = reinterpret_cast<float4 *>(out) + (size_t)row * 3 * Q;
  float mx = 0.f;
  for (int c = l; c < Q; c += 32) {
    const float4 x = ar[c], y = br[c];
    const float4 p = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
    o[c] = x;
    o[Q + c] = y;
    o[2 * Q + c] = p;
    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))),
                         fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)))));
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fmaxf(fabsf(p.z), fabsf(p.w))));
  }
  mx = group_max<32>(mx);
  if (l == 0) rowmax[row] = mx;
}

__global__ void node_to_edge_mask_kernel(const float *__restrict__ mask, const int64_t *__restrict__ ei, int64_t E,
                                         float *__restrict__ out) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  out[e] = mask[ei[e]] * mask[ei[E + e]];
}

// ---- scatter-mean over the CSR plan: one wave per destination node --------------------------------
__global__ __launch_bounds__(256) void scatter_mean_kernel(const float4 *__restrict__ msg, const int *__restrict__ rowptr,
                                                           const int *__restrict__ eid, float4 *__restrict__ out, int N,
                                                           int Q) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int rb = rowptr[i], re = rowptr[i + 1];
  const float cnt = (float)max(re - rb, 1);
  for (int q = lane; q < Q; q += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = rb; t < re; ++t) {
      float4 v = msg[(size_t)eid[t] * Q + q];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt;
    out[(size_t)i * Q + q] = acc;
  }
}

}  // namespace isg

using namespace isg;

static int mp_fwd(const void *x_l, const void *x_r, const void *e_proj, const float *att, const float *bias,
                  const int32_t *rowptr, const int32_t *eid, const int32_t *src, const float *node_mask,
                  const float *edge_mask, void *out, float *alpha, int64_t N, int64_t E, int32_t H, int32_t C,
                  float negative_slope, const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst,
                  int64_t B, int32_t nmax_host, int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e,
                  void *stream, int f16, float *rowmax = nullptr, const float *logits = nullptr, uint16_t *planes = nullptr,
                  float *planes_inv = nullptr) {
  if (N < 0 || E < 0 || H <= 0 || C <= 0) return ISG_EINVAL;
  if (N == 0) return ISG_OK;
  if (!x_l || !att || !rowptr || (!out && !planes) || (E > 0 && (!eid || !src || !alpha))) return ISG_EINVAL;
  if (!logits && (!x_r || (E > 0 && !e_proj))) return ISG_EINVAL;      // with logits handed in, e_proj and x_r are not read
  if ((C & 3) != 0 || N >= (1ll << 31) || E >= (1ll << 31)) return ISG_EUNSUPPORTED;
  if (ld_l == 0) ld_l = H * C;
  if (ld_r == 0) ld_r = H * C;
  if (ld_e == 0) ld_e = H * C;
  if (ld_l < H * C || ld_r < H * C || ld_e < H * C) return ISG_EINVAL;
  if ((ld_l & 3) != 0 || (ld_r & 3) != 0 || (ld_e & 3) != 0) return ISG_EUNSUPPORTED;
  static const int mp_flags = [] { const char *f = getenv("ISG_MP_FLAGS"); return f ? atoi(f) : ISG_MP_DEFAULT_FLAGS; }();
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  MpArgs a = {
      .x_l = (const float4 *)x_l, .x_r = (const float4 *)x_r, .e_proj = (const float4 *)e_proj, .att = (const float4 *)att,
      .bias = (const float4 *)bias, .rowptr = rowptr, .eid = eid, .src = src, .node_mask = node_mask, .edge_mask = edge_mask,
      .out = (float4 *)out, .alpha = alpha, .N = (int)N, .C = C, .H = H, .lde4 = ld_e >> 2, .ldl4 = ld_l >> 2, .ldr4 = ld_r >> 2,
      .slope = negative_slope, .graph_ptr = graph_ptr, .graph_eptr = graph_eptr, .dst = dst, .B = (int)B, .lrows = 0, .f16 = f16,
      .flags = mp_flags,           // experiment switch, read once; default = tuned setting
      .nchunks = 0, .logits = logits, .rowmax = rowmax, .planes = planes, .planes_inv = planes_inv,
      .planes_kt = 2 * ((2 * (C >> 2) + 7) >> 3)};     // two half rows of 2 C columns, each padded to whole 32-column lines
  if (!a.x_l || !a.att || !a.rowptr || (!a.out && !a.planes) || (E > 0 && (!a.eid || !a.src || !a.alpha)) ||
      (!a.logits && (!a.x_r || (E > 0 && !a.e_proj))) || (a.planes && !a.planes_inv))
    return ISG_EINVAL;                         // the struct the kernels dereference, not the parameters it was filled from
  hipStream_t st = as_stream(stream);
  if (graph_ptr && graph_eptr && (dst || E == 0) && B > 0 && B < (1ll << 31) && nmax_host > 0) {
    int rc = launch_mp_graph(a, nmax_host, emax_host, st);
    if (rc != ISG_EUNSUPPORTED) return rc;   // shapes without a per-graph instantiation use the node-chunk kernel
  }
  if (rowmax || logits || planes) return ISG_EUNSUPPORTED;   // row maxima out / logits in / planes out: per-graph kernels only
  if (f16) return ISG_EUNSUPPORTED;          // fp16 rows exist in the per-graph kernel only
  switch (H) {
    case 1: return launch_mp<1>(a, st);
    case 2: return launch_mp<2>(a, st);
    case 4: return launch_mp<4>(a, st);
    case 8: return launch_mp<8>(a, st);
    default: return ISG_EUNSUPPORTED;
  }
}

extern "C" int isg_gatv2_mp_fwd(const float *x_l, const float *x_r, const float *e_proj, const float *att,
                                const float *bias, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                const float *node_mask, const float *edge_mask, float *out, float *alpha, int64_t N,
                                int64_t E, int32_t H, int32_t C, float negative_slope, const int32_t *graph_ptr,
                                const int32_t *graph_eptr, const int32_t *dst, int64_t B, int32_t nmax_host,
                                int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e, void *stream) {
  return mp_fwd(x_l, x_r, e_proj, att, bias, rowptr, eid, src, node_mask, edge_mask, out, alpha, N, E, H, C,
                negative_slope, graph_ptr, graph_eptr, dst, B, nmax_host, emax_host, ld_l, ld_r, ld_e, stream, 0);
}

extern "C" int isg_gatv2_mp_fwd_f16(const uint16_t *x_l, const uint16_t *x_r, const uint16_t *e_proj, const float *att,
                                    const float *bias, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                    const float *node_mask, const float *edge_mask, uint16_t *out, float *alpha,
                                    int64_t N, int64_t E, int32_t H, int32_t C, float negative_slope,
                                    const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst, int64_t B,
                                    int32_t nmax_host, int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e,
                                    void *stream) {
  return mp_fwd(x_l, x_r, e_proj, att, bias, rowptr, eid, src, node_mask, edge_mask, out, alpha, N, E, H, C,
                negative_slope, graph_ptr, graph_eptr, dst, B, nmax_host, emax_host, ld_l, ld_r, ld_e, stream, 1);
}

extern "C" int isg_instr_gate(const float *x, const float *instr, const int64_t *batch, float *out, int64_t N,
                              int32_t C, void *stream) {
  if (N < 0 || C <= 0) return ISG_EINVAL;
  if (N == 0) return ISG_OK;
  if (!x || !instr || !batch || !out) return ISG_EINVAL;
  if ((C & 3) != 0) return ISG_EUNSUPPORTED;
  const int Q = C >> 2;
  const int64_t total = N * Q;
  if ((total + 255) / 256 >= (1ll << 31)) return ISG_EUNSUPPORTED;
  instr_gate_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(
      (const float4 *)x, (const float4 *)instr, batch, (float4 *)out, total, Q);
  return check_launch();
}

extern "C" int isg_cat_mul_rowmax(const float *a, const float *b, float *out, float *rowmax, int64_t M, int32_t C, void *stream) {
  if (M < 0 || C <= 0) return ISG_EINVAL;
  if ((C & 3) != 0 || M >= (1ll << 31) || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                                            reinterpret_cast<uintptr_t>(out)) & 15) != 0)
    return ISG_EUNSUPPORTED;
  if (M == 0) return ISG_OK;
  if (!a || !b || !out || !rowmax) return ISG_EINVAL;
  cat_mul_rowmax_kernel<<<(unsigned)((M + 7) / 8), 256, 0, as_stream(stream)>>>(a, b, out, rowmax, (int)M, C >> 2);
  return check_launch();
}

extern "C" int isg_node_to_edge_mask(const float *node_mask, const int64_t *edge_index, int64_t E, float *out,
                                     void *stream) {
  if (E < 0) return ISG_EINVAL;
  if (E == 0) return ISG_OK;
  if (!node_mask || !edge_index || !out) return ISG_EINVAL;
  node_to_edge_mask_kernel<<<(unsigned)((E + 255) / 256), 256, 0, as_stream(stream)>>>(node_mask, edge_index, E, out);
  return check_launch();
}

extern "C" int isg_scatter_mean(const float *msg, const int32_t *rowptr, const int32_t *eid, float *out, int64_t N,
                                int32_t C, void *stream) {
  if (N < 0 || C <= 0) return ISG_EINVAL;
  if (N == 0) return ISG_OK;
  if (!rowptr || !out) return ISG_EINVAL;
  if ((C & 3) != 0 || N >= (1ll << 31)) return ISG_EUNSUPPORTED;
  scatter_mean_kernel<<<(unsigned)((N + 3) / 4), 256, 0, as_stream(stream)>>>(
      (const float4 *)msg, rowptr, eid, (float4 *)out, (int)N, C >> 2);
  return check_launch();
}

// isg_gatv2_mp_fwd that also writes rowmax[N, H] = max |out[n, h*C : (h+1)*C]| (per-graph grouped kernel only;
// ISG_EUNSUPPORTED for shapes that would take another kernel: the caller then runs isg_gatv2_mp_fwd)
extern "C" int isg_gatv2_mp_fwd_rowmax(const float *x_l, const float *x_r, const float *e_proj, const float *att,
                                       const float *bias, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                       const float *node_mask, const float *edge_mask, float *out, float *alpha,
                                       float *rowmax, int64_t N, int64_t E, int32_t H, int32_t C, float negative_slope,
                                       const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst, int64_t B,
                                       int32_t nmax_host, int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e,
                                       void *stream) {
  if (!rowmax) return ISG_EINVAL;
  return mp_fwd(x_l, x_r, e_proj, att, bias, rowptr, eid, src, node_mask, edge_mask, out, alpha, N, E, H, C,
                negative_slope, graph_ptr, graph_eptr, dst, B, nmax_host, emax_host, ld_l, ld_r, ld_e, stream, 0, rowmax);
}

// isg_gatv2_mp_fwd whose result leaves as the SEGMENTED planes32 operand of isg_linear_h3p instead of fp32 rows (H = 4, the flat
// per-graph kernel of head dimensions like the reference's C = 300; ISG_EUNSUPPORTED wherever that kernel does not run -- the
// caller then takes isg_gatv2_mp_fwd and isg_split_planes32): out_planes uint16 [N][2 * ceil(2 C / 32)][64], out_inv fp32 [2][N]
// = the row scales of columns [0, 2C) and [2C, 4C).  x_proj.0 (mgat.py:156) reads it with no pass in between.
extern "C" int isg_gatv2_mp_fwd_planes(const float *x_l, const float *x_r, const float *e_proj, const float *att,
                                       const float *bias, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                       const float *node_mask, const float *edge_mask, uint16_t *out_planes, float *out_inv,
                                       float *alpha, int64_t N, int64_t E, int32_t H, int32_t C, float negative_slope,
                                       const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst, int64_t B,
                                       int32_t nmax_host, int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e,
                                       void *stream) {
  if (!out_planes || !out_inv) return ISG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(out_planes) & 15) != 0 || H != 4) return ISG_EUNSUPPORTED;
  return mp_fwd(x_l, x_r, e_proj, att, bias, rowptr, eid, src, node_mask, edge_mask, nullptr, alpha, N, E, H, C,
                negative_slope, graph_ptr, graph_eptr, dst, B, nmax_host, emax_host, ld_l, ld_r, ld_e, stream, 0, nullptr,
                nullptr, out_planes, out_inv);
}

// isg_gatv2_mp_fwd with the attention logits handed in (fp32 [E, H] in CSR slot order, from isg_gatv2_edge_logits)
// instead of e_proj and x_r: masked softmax over every destination's in-edges, alpha out, aggregation of x_l rows, bias,
// optional row maxima of `out`.  Per-graph grouped kernel only (ISG_EUNSUPPORTED otherwise: the caller then runs the
// un-fused pair).
extern "C" int isg_gatv2_mp_fwd_logits(const float *x_l, const float *logits, const float *att, const float *bias,
                                       const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                       const float *node_mask, const float *edge_mask, float *out, float *alpha,
                                       float *rowmax, int64_t N, int64_t E, int32_t H, int32_t C, float negative_slope,
                                       const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst, int64_t B,
                                       int32_t nmax_host, int32_t emax_host, int32_t ld_l, void *stream) {
  if (!logits && E > 0) return ISG_EINVAL;
  if (E == 0) return ISG_EUNSUPPORTED;
  return mp_fwd(x_l, nullptr, nullptr, att, bias, rowptr, eid, src, node_mask, edge_mask, out, alpha, N, E, H, C,
                negative_slope, graph_ptr, graph_eptr, dst, B, nmax_host, emax_host, ld_l, 0, 0, stream, 0, rowmax, logits);
}

// isg_gatv2_mp_fwd_logits with HALF feature rows (x_l in, out; ld_l in halves): BASELINE configs[4]'s storage behind
// isg_gatv2_edge_logits_f16.  The grouped per-graph kernel only.
extern "C" int isg_gatv2_mp_fwd_logits_f16(const uint16_t *x_l, const float *logits, const float *att, const float *bias,
                                           const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                           const float *node_mask, const float *edge_mask, uint16_t *out, float *alpha,
                                           int64_t N, int64_t E, int32_t H, int32_t C, float negative_slope,
                                           const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst, int64_t B,
                                           int32_t nmax_host, int32_t emax_host, int32_t ld_l, void *stream) {
  if (!logits && E > 0) return ISG_EINVAL;
  if (E == 0) return ISG_EUNSUPPORTED;
  return mp_fwd(x_l, nullptr, nullptr, att, bias, rowptr, eid, src, node_mask, edge_mask, out, alpha, N, E, H, C,
                negative_slope, graph_ptr, graph_eptr, dst, B, nmax_host, emax_host, ld_l, 0, 0, stream, 1, nullptr, logits);
}

// isg_gatv2_mp_fwd_logits whose result leaves as the SEGMENTED planes32 operand of isg_linear_h3p (isg_gatv2_mp_fwd_planes's
// output contract: H = 4, the flat per-graph kernel): the C = 300 layer without e_proj AND without fp32 conv rows.
extern "C" int isg_gatv2_mp_fwd_logits_planes(const float *x_l, const float *logits, const float *att, const float *bias,
                                              const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                              const float *node_mask, const float *edge_mask, uint16_t *out_planes,
                                              float *out_inv, float *alpha, int64_t N, int64_t E, int32_t H, int32_t C,
                                              float negative_slope, const int32_t *graph_ptr, const int32_t *graph_eptr,
                                              const int32_t *dst, int64_t B, int32_t nmax_host, int32_t emax_host,
                                              int32_t ld_l, void *stream) {
  if ((!logits && E > 0) || !out_planes || !out_inv) return ISG_EINVAL;
  if (E == 0 || (reinterpret_cast<uintptr_t>(out_planes) & 15) != 0 || H != 4) return ISG_EUNSUPPORTED;
  return mp_fwd(x_l, nullptr, nullptr, att, bias, rowptr, eid, src, node_mask, edge_mask, nullptr, alpha, N, E, H, C,
                negative_slope, graph_ptr, graph_eptr, dst, B, nmax_host, emax_host, ld_l, 0, 0, stream, 0, nullptr, logits,
                out_planes, out_inv);
}
