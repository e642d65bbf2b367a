// Backward of the GATv2 message-passing operator (training of the hot path).
//
// Forward (isg_gatv2_mp_fwd), per target i, head h, incoming edge e = (j -> i), mask m_e:
//   s = x_r[i] + x_l[j] + ep[e];  u = leaky(s*m)*m;  a_e = <u, att_h>;  alpha = softmax_i(a);  out_i = sum_e alpha_e m_e x_l[j]
// Given g = d out_i:
//   dalpha_e = m_e <g, x_l[j]>
//   da_e     = alpha_e (dalpha_e - sum_e' alpha_e' dalpha_e')                      (softmax; the +1e-16 is below fp32 resolution)
//   ds_e     = da_e * att_h * m_e^2 * (s*m > 0 ? 1 : slope)
//   d att_h += da_e * u_e          d x_r[i] += ds_e          d ep[e] = ds_e
//   d x_l[j] += ds_e + alpha_e m_e g                                               (a scatter by SOURCE)
//   d m_e     = sum_h [ da_e <att_h, L'(s m) s m + L(s m)> + alpha_e <g, x_l[j]> ]   (optional: the edge-mask gradient
//               the I-MLE / AIMLE / straight-through samplers are trained through)
// Two launches, both without atomics (fixed summation order, bitwise reproducible):
//   K1  destination-major (the node-chunk decomposition of isg_mp.hip: workgroup = 16 targets, wave = one target, H lane
//       groups): writes d ep, d x_r and one partial row of d att per workgroup;
//   K2  source-major over a CSR-by-source plan: d x_l[j] = sum over j's out-edges, in edge-id order, of
//       (d ep[e] + alpha_e m_e g[dst_e]).
// The gradient flowing into the returned attention weights `alpha` is not supported (the model never consumes them).
#include "isg_mp.hpp"

namespace isg {

struct MpBwdArgs {
  const float4 *x_l, *x_r, *e_proj, *att, *grad_out;
  const float *alpha;
  const int *rowptr, *eid, *src;           // CSR by destination
  const int *rowptr_s, *eid_s, *dst_s;     // CSR by source (isg_csr_build on the flipped edge_index)
  const float *node_mask, *edge_mask;      // optional (NULL: the layer is not masked)
  float4 *d_e_proj, *d_x_r, *d_x_l, *d_att_partial;
  float *d_edge_mask;                      // optional [E]
  int N, C, H;
  float slope;
};

template <int H, int P>
__global__ __launch_bounds__(MP_WAVES * 64) void gatv2_mp_bwd_dst_kernel(MpBwdArgs a) {
  constexpr int G = 64 / H;
  __shared__ int s_rowptr[MP_NPB + 1];
  __shared__ int s_src[MP_ECAP];
  __shared__ int s_eid[MP_ECAP];
  __shared__ float s_da[MP_WAVES][MP_LCAP * H * 2];   // {m_e <g,x_l>, <g,x_l>} per slot and head
  __shared__ float4 s_datt[MP_WAVES][64 * P];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * MP_NPB;
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
  const int Q = a.C >> 2, R = H * Q;
  int off[P];
  bool ok[P];
  float4 att4[P], datt[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int q = p * G + l;
    ok[p] = q < Q;
    off[p] = g * Q + (ok[p] ? q : 0);
    att4[p] = ok[p] ? a.att[off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
    datt[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int mode = a.edge_mask ? 2 : (a.node_mask ? 1 : 0);
  float *s_w = s_da[wave];

  for (int k = wave; k < nn; k += MP_WAVES) {
    const int i = n0 + k;
    const int rb = s_rowptr[k], re = s_rowptr[k + 1];
    float4 xr4[P], g4[P], dxr[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      xr4[p] = ok[p] ? a.x_r[(size_t)i * R + off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
      g4[p] = ok[p] ? a.grad_out[(size_t)i * R + off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
      dxr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float mi = mode == 1 ? a.node_mask[i] : 1.f;

    // pass 1: dalpha_e = m_e <g, x_l[j]> and S = sum alpha_e dalpha_e (edge order)
    float S = 0.f;
    for (int t = rb; t < re; ++t) {
      const int rel = t - e0;
      int j, e;
      if (rel < MP_ECAP) { j = s_src[rel]; e = s_eid[rel]; } else { j = a.src[t]; e = a.eid[t]; }
      float me = 1.f;
      if (mode == 1) me = a.node_mask[j] * mi;
      else if (mode == 2) me = a.edge_mask[e];
      const float4 *xl = a.x_l + (size_t)j * R;
      float part = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (ok[p]) part += dot4(g4[p], xl[off[p]]);
      const float raw = group_sum<G>(part);
      const float dal = raw * me;
      S += a.alpha[(size_t)e * H + g] * dal;
      const int slot = t - rb;
      if (l == 0) {
        if (slot < MP_LCAP) {
          s_w[(slot * H + g) * 2] = dal;
          s_w[(slot * H + g) * 2 + 1] = raw;
        } else {   // parked in the d_e_proj row that pass 2 overwrites
          float *park = reinterpret_cast<float *>(a.d_e_proj) + (size_t)e * H * a.C + g * a.C;
          park[0] = dal;
          park[1] = raw;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();

    // pass 2: da_e, then everything that flows through the logit
    for (int t = rb; t < re; ++t) {
      const int rel = t - e0;
      int j, e;
      if (rel < MP_ECAP) { j = s_src[rel]; e = s_eid[rel]; } else { j = a.src[t]; e = a.eid[t]; }
      float me = 1.f;
      if (mode == 1) me = a.node_mask[j] * mi;
      else if (mode == 2) me = a.edge_mask[e];
      const int slot = t - rb;
      float dal, raw;
      if (slot < MP_LCAP) {
        dal = s_w[(slot * H + g) * 2];
        raw = s_w[(slot * H + g) * 2 + 1];
      } else {
        const float *park = reinterpret_cast<const float *>(a.d_e_proj) + (size_t)e * H * a.C + g * a.C;
        dal = l == 0 ? park[0] : 0.f;
        raw = l == 0 ? park[1] : 0.f;
        dal = __shfl(dal, lane - l, 64);
        raw = __shfl(raw, lane - l, 64);
      }
      const float al = a.alpha[(size_t)e * H + g];
      const float da = al * (dal - S);
      float dm_part = 0.f;
      const float4 *xl = a.x_l + (size_t)j * R;
      const float4 *ep = a.e_proj + (size_t)e * R;
      float4 *dep = a.d_e_proj + (size_t)e * R;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          const float4 u = xl[off[p]], v = ep[off[p]];
          float s[4] = {(xr4[p].x + u.x) + v.x, (xr4[p].y + u.y) + v.y, (xr4[p].z + u.z) + v.z, (xr4[p].w + u.w) + v.w};
          const float at[4] = {att4[p].x, att4[p].y, att4[p].z, att4[p].w};
          float ds[4], uu[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float sm = mode != 0 ? s[c] * me : s[c];
            const float lk = sm > 0.f ? sm : sm * a.slope;
            uu[c] = mode != 0 ? lk * me : lk;
            const float dlk = sm > 0.f ? 1.f : a.slope;
            ds[c] = da * at[c] * dlk;
            if (mode != 0) ds[c] *= me * me;
            dm_part += at[c] * (dlk * s[c] * me + lk);
          }
          datt[p].x += da * uu[0]; datt[p].y += da * uu[1]; datt[p].z += da * uu[2]; datt[p].w += da * uu[3];
          dxr[p].x += ds[0]; dxr[p].y += ds[1]; dxr[p].z += ds[2]; dxr[p].w += ds[3];
          dep[off[p]] = make_float4(ds[0], ds[1], ds[2], ds[3]);
        }
      }
      if (a.d_edge_mask) {   // sum over channels (group), then over heads (one lane per group contributes)
        const float dmh = da * group_sum<G>(dm_part) + al * raw;
        const float tot = wave_sum(l == 0 ? dmh : 0.f);
        if (lane == 0) a.d_edge_mask[e] = tot;
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (ok[p]) a.d_x_r[(size_t)i * R + off[p]] = dxr[p];
    __builtin_amdgcn_wave_barrier();
  }

  // d att: waves -> LDS -> one partial row per workgroup (summed over workgroups by the caller, fixed order)
#pragma unroll
  for (int p = 0; p < P; ++p) s_datt[wave][p * 64 + lane] = datt[p];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (ok[p]) {
        float4 t = s_datt[0][p * 64 + lane];
        for (int w = 1; w < MP_WAVES; ++w) {
          const float4 o = s_datt[w][p * 64 + lane];
          t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        a.d_att_partial[(size_t)blockIdx.x * R + off[p]] = t;
      }
    }
  }
}

// K2: one wave per source node; d x_l[j] = sum_{e in out(j)} (d ep[e] + alpha_e m_e g[dst_e]), ascending edge id
template <int H, int P>
__global__ __launch_bounds__(256) void gatv2_mp_bwd_src_kernel(MpBwdArgs a) {
  constexpr int G = 64 / H;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= a.N) return;
  const int g = lane / G, l = lane % G;
  const int Q = a.C >> 2, R = H * Q;
  int off[P];
  bool ok[P];
  float4 acc[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int q = p * G + l;
    ok[p] = q < Q;
    off[p] = g * Q + (ok[p] ? q : 0);
    acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int mode = a.edge_mask ? 2 : (a.node_mask ? 1 : 0);
  const float mj = mode == 1 ? a.node_mask[j] : 1.f;
  const int rb = a.rowptr_s[j], re = a.rowptr_s[j + 1];
  for (int t = rb; t < re; ++t) {
    const int e = a.eid_s[t], d = a.dst_s[t];
    float me = 1.f;
    if (mode == 1) me = mj * a.node_mask[d];
    else if (mode == 2) me = a.edge_mask[e];
    const float w = a.alpha[(size_t)e * H + g] * me;
    const float4 *dep = a.d_e_proj + (size_t)e * R;
    const float4 *go = a.grad_out + (size_t)d * R;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (ok[p]) {
        const float4 x = dep[off[p]], y = go[off[p]];
        acc[p].x += x.x + w * y.x;
        acc[p].y += x.y + w * y.y;
        acc[p].z += x.z + w * y.z;
        acc[p].w += x.w + w * y.w;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (ok[p]) a.d_x_l[(size_t)j * R + off[p]] = acc[p];
}

// NodeMaskToEdgeMask.backward (ISubGVQA/sampling/node_edge_masks.py:13-19): the reference scatters the edge-mask
// gradient to the DESTINATION node only (no product rule); reproduced as is, in edge-id order.
__global__ void node_mask_bwd_kernel(const float *__restrict__ d_edge, const int *__restrict__ rowptr,
                                     const int *__restrict__ eid, float *__restrict__ d_node, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f;
  for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) acc += d_edge[eid[t]];
  d_node[i] = acc;
}

template <int H>
static int launch_bwd(const MpBwdArgs &a, hipStream_t st) {
  constexpr int G = 64 / H;
  const int Q = a.C >> 2;
  const int P = (Q + G - 1) / G;
  const int blocks = (a.N + MP_NPB - 1) / MP_NPB;
#define ISG_BW(p)                                                                              \
  case p:                                                                                      \
    gatv2_mp_bwd_dst_kernel<H, p><<<blocks, MP_WAVES * 64, 0, st>>>(a);                        \
    gatv2_mp_bwd_src_kernel<H, p><<<(a.N + 3) / 4, 256, 0, st>>>(a);                           \
    break;
  switch (P) {
    ISG_BW(1) ISG_BW(2) ISG_BW(3) ISG_BW(4) ISG_BW(5) ISG_BW(6) ISG_BW(7) ISG_BW(8)
    default: return ISG_EUNSUPPORTED;
  }
#undef ISG_BW
  return check_launch();
}

}  // namespace isg

using namespace isg;

extern "C" int isg_gatv2_mp_bwd(const float *x_l, const float *x_r, const float *e_proj, const float *att,
                                const float *alpha, const float *grad_out, const int32_t *rowptr, const int32_t *eid,
                                const int32_t *src, const int32_t *rowptr_s, const int32_t *eid_s, const int32_t *dst_s,
                                const float *node_mask, const float *edge_mask, float *d_x_l, float *d_x_r,
                                float *d_e_proj, float *d_att_partial, float *d_edge_mask, int64_t N, int64_t E,
                                int32_t H, int32_t C, float negative_slope, void *stream) {
  if (N < 0 || E < 0 || H <= 0 || C <= 0) return ISG_EINVAL;
  if (N == 0) return ISG_OK;
  if (!x_l || !x_r || !att || !grad_out || !rowptr || !rowptr_s || !d_x_l || !d_x_r || !d_att_partial) return ISG_EINVAL;
  if (E > 0 && (!e_proj || !alpha || !eid || !src || !eid_s || !dst_s || !d_e_proj)) return ISG_EINVAL;
  if ((C & 3) != 0 || N >= (1ll << 31) || E >= (1ll << 31)) return ISG_EUNSUPPORTED;
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  MpBwdArgs a = {
      .x_l = (const float4 *)x_l, .x_r = (const float4 *)x_r, .e_proj = (const float4 *)e_proj, .att = (const float4 *)att,
      .grad_out = (const float4 *)grad_out, .alpha = alpha, .rowptr = rowptr, .eid = eid, .src = src, .rowptr_s = rowptr_s,
      .eid_s = eid_s, .dst_s = dst_s, .node_mask = node_mask, .edge_mask = edge_mask, .d_e_proj = (float4 *)d_e_proj,
      .d_x_r = (float4 *)d_x_r, .d_x_l = (float4 *)d_x_l, .d_att_partial = (float4 *)d_att_partial, .d_edge_mask = d_edge_mask,
      .N = (int)N, .C = C, .H = H, .slope = negative_slope};
  if (!a.x_l || !a.x_r || !a.att || !a.grad_out || !a.rowptr || !a.rowptr_s || !a.d_x_l || !a.d_x_r || !a.d_att_partial ||
      (E > 0 && (!a.e_proj || !a.alpha || !a.eid || !a.src || !a.eid_s || !a.dst_s || !a.d_e_proj)))
    return ISG_EINVAL;                         // the struct the kernels dereference, not the parameters it was filled from
  hipStream_t st = as_stream(stream);
  switch (H) {
    case 1: return launch_bwd<1>(a, st);
    case 2: return launch_bwd<2>(a, st);
    case 4: return launch_bwd<4>(a, st);
    case 8: return launch_bwd<8>(a, st);
    default: return ISG_EUNSUPPORTED;
  }
}

extern "C" int isg_node_to_edge_mask_bwd(const float *d_edge_mask, const int32_t *rowptr, const int32_t *eid,
                                         float *d_node_mask, int64_t N, void *stream) {
  if (N < 0) return ISG_EINVAL;
  if (N == 0) return ISG_OK;
  if (!d_edge_mask || !rowptr || !eid || !d_node_mask) return ISG_EINVAL;
  if (N >= (1ll << 31)) return ISG_EUNSUPPORTED;
  node_mask_bwd_kernel<<<(unsigned)((N + 255) / 256), 256, 0, as_stream(stream)>>>(d_edge_mask, rowptr, eid, d_node_mask, (int)N);
  return check_launch();
}
