// GATv2 message passing with the edge projection computed INSIDE the kernel: e_proj = lin_edge(edge_attr) never exists in
// HBM (reference: ISubGVQA/models/mgat_v2_conv.py:259-261 inside message(), :243-279).
//
// Un-fused, lin_edge writes [E, H*C] fp32 (420 MB at configs[1]) that the message-passing kernel reads back once: 45 % of
// that kernel's traffic and the whole output of a GEMM that runs at the chip's write rate.  Here the per-graph kernel of
// isg_mp_graph.hip (same phases, tables, arithmetic and summation orders for everything but the logit reduction) gets
// its e_proj tile from the matrix cores:
//
//   workgroup  one graph x one head pair (256 output columns), 8 waves, 2 workgroups per CU
//   phase A    x_l slice, CSR records, row pointers -> LDS                                   (as isg_mp_graph.hip)
//   phase B    per block of 64 CSR slots:
//     B0  the slots' edge_attr rows (gathered by original edge id) -> three bf16 planes in LDS, one 64-wide k half at a time
//     B1  wave w = 32-column subtile w of the head pair: [64 edges x K] . W_e[subtile]^T on v_mfma_f32_32x32x16_bf16, six
//         terms per product (fp32-level accuracy, isg_gemm_panel.hip), W fragments straight from L2 (fragment-major planes)
//     B2  logits out of the accumulator layout (lane = channel, register = edge): s = (x_r[dst] + x_l[src]) + e_proj,
//         mask / leaky_relu / mask, x att, 32-lane sum -> one partial per (edge, subtile) in LDS
//     B3  the four partials of a head are added in subtile order -> logit table
//   phase C    node-parallel softmax + aggregation from LDS                                  (as isg_mp_graph.hip)
// Restrictions (anything else takes the un-fused path): fp32 rows, C = 128, H even, edge features K <= 128 with 4 | K,
// graphs within the 64-node / 256-edge tables.
#include "isg_mp.hpp"

namespace isg {

typedef __attribute__((ext_vector_type(8))) __bf16 fbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 fbf16x4;
typedef __attribute__((ext_vector_type(16))) float ff32x16;
typedef int fe_i32x4 __attribute__((ext_vector_type(4)));

constexpr int FE_WAVES = 8, FE_THREADS = 512;
constexpr int FE_NCAP = 64, FE_ECAP = 256;   // nodes / CSR slots per graph (isg_mp_graph.hip's small tables)
constexpr int FE_EB = 64;                    // CSR slots per MFMA block (two 32-row tiles)
constexpr int FE_KH = 64;                    // k per staged half
constexpr int FE_LD = FE_KH + 8;             // bf16 per LDS row (144 B: conflict-free 16-byte fragment reads)
constexpr int FE_HS = 2, FE_C = 128;         // heads per workgroup, channels per head
constexpr int FE_RQ = FE_HS * FE_C / 4;      // float4 per staged x_l row slice (64)

struct FeArgs {
  MpArgs m;                 // e_proj unused
  const float *edge_attr;   // [E, K] fp32, row stride ld_ea floats
  const __bf16 *w_frag;     // fragment-major planes of lin_edge.weight [H*C, K] (isg_split_bf16x3_frag)
  int K, KS, NT, ld_ea;
};

__device__ __forceinline__ int fe_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float fe_bf16_to_f32(__bf16 v) {
  return __uint_as_float(((unsigned)__builtin_bit_cast(unsigned short, v)) << 16);
}
__device__ __forceinline__ void fe_split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
  p1 = (__bf16)x;
  const float r1 = x - fe_bf16_to_f32(p1);
  p2 = (__bf16)r1;
  const float r2 = r1 - fe_bf16_to_f32(p2);
  p3 = (__bf16)r2;
}

template <bool MASKED>
__global__ __launch_bounds__(FE_THREADS, 4) void gatv2_mp_fused_edge_kernel(FeArgs fa) {
  const MpArgs &a = fa.m;
  extern __shared__ __attribute__((aligned(16))) float4 s_xl[];            // [lrows][FE_RQ]
  __shared__ __attribute__((aligned(16))) __bf16 sA[3][FE_EB][FE_LD];      // 27,648 B
  __shared__ __attribute__((aligned(16))) int4 s_tab[FE_ECAP];             // {src - nb, eid, dst - nb, bits(edge mask)}
  __shared__ __attribute__((aligned(16))) float s_lg[FE_ECAP * FE_HS];
  __shared__ __attribute__((aligned(16))) float s_part[FE_ECAP * FE_WAVES];
  __shared__ __attribute__((aligned(16))) int s_rowptr[FE_NCAP + 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = fe_uni(tid >> 6);
  const int nhg = a.H / FE_HS;
  const int item = blockIdx.x;
  const int g = item / nhg, hg = item - g * nhg;
  const int nb = a.graph_ptr[g], n = min(a.graph_ptr[g + 1] - nb, FE_NCAP);
  const int e0 = a.graph_eptr[g], ne = min(a.graph_eptr[g + 1] - e0, FE_ECAP);
  if (n <= 0) return;
  const int R = a.H * (FE_C / 4);     // float4 per full row
  const int hoff = hg * FE_RQ;        // first float4 of this head pair inside a row
  const int rows = min(n, a.lrows);
  const bool nt = a.flags & 1, nt_xl = a.flags & 4;

  // Everything a block needs from global memory is REQUESTED before anything is waited for: the first 64 slots' edge_attr
  // rows (both k halves; the slot -> edge id map is read straight from the CSR arrays, not through the LDS records) go out
  // together with phase A's loads, the W fragments run one k-step ahead, and the x_r values of the logit phase are
  // requested before the second half's MFMAs.  A workgroup is a short serial chain (32 of them per CU, two at a time):
  // the first version, which loaded each operand where it needed it, spent 466 us against 394 us for the un-fused pair.
  const int fr = lane & 31, fk = (lane >> 5) * 8, h = lane >> 5;
  const int hh = wave >> 2;                                  // head of this wave's subtile inside the pair
  const int ccol = hh * FE_C + (wave & 3) * 32 + fr;         // this lane's column inside the 256-column slice
  const int nhalf = (fa.K + FE_KH - 1) / FE_KH;
  float4 av[2][2];                                           // [u][k half]: staging image of the current block
#define FE_LOAD_BLOCK(eb_, eid_of)                                                                               \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                \
    const int i = tid + FE_THREADS * u;                                                                          \
    const int row = i >> 4, c4 = i & 15;                                                                         \
    const int e = eid_of(min((eb_) + row, ne - 1));                                                              \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) {                                                           \
      const int gk = min(kh * FE_KH + c4 * 4, fa.K - 4);                                                         \
      av[u][kh] = *reinterpret_cast<const float4 *>(fa.edge_attr + (size_t)e * fa.ld_ea + gk);                   \
    }                                                                                                            \
  }
#define FE_EID_GLOBAL(t) a.eid[e0 + (t)]
#define FE_EID_LDS(t) s_tab[(t)].y
  if (ne > 0) FE_LOAD_BLOCK(0, FE_EID_GLOBAL)

  // ---- phase A (isg_mp_graph.hip) -------------------------------------------------------------------------------------
  {
    int4 rec = make_int4(0, 0, 0, __float_as_int(1.f));
    int v_rp = 0;
    if (tid <= n) v_rp = a.rowptr[nb + tid] - e0;
    if (tid < ne) {
      const int s = a.src[e0 + tid], e = a.eid[e0 + tid], d = a.dst[e0 + tid];
      rec.x = min(max(s - nb, 0), n - 1);
      rec.y = e;
      rec.z = min(max(d - nb, 0), n - 1);
      if (MASKED) {
        float me;
        if (a.edge_mask) me = a.edge_mask[e];
        else me = a.node_mask[s] * a.node_mask[d];
        rec.w = __float_as_int(me);
      }
    }
#pragma unroll 1
    for (int r0 = wave; r0 < rows; r0 += FE_WAVES * 2) {    // FE_RQ = 64 float4 per row: one per lane
      const int r1 = r0 + FE_WAVES;
      float4 v0 = ld_stream(a.x_l + (size_t)(nb + r0) * a.ldl4 + hoff + lane, nt_xl), v1;
      if (r1 < rows) v1 = ld_stream(a.x_l + (size_t)(nb + r1) * a.ldl4 + hoff + lane, nt_xl);
      s_xl[r0 * FE_RQ + lane] = v0;
      if (r1 < rows) s_xl[r1 * FE_RQ + lane] = v1;
    }
    if (tid <= n) s_rowptr[tid] = v_rp;
    if (tid < ne) s_tab[tid] = rec;
  }

  // ---- phase B: e_proj on the matrix cores, logits out of the accumulators ---------------------------------------------
  const float attv = reinterpret_cast<const float *>(a.att)[hoff * 4 + ccol];
  const float slope = a.slope;
  const unsigned plane_b = (unsigned)fa.NT * (unsigned)fa.KS * 1024u;
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(fa.w_frag), 0, (int)(3u * plane_b), 0x00020000);
  const int voff = lane * 16;
  const unsigned wb = (unsigned)(hg * 8 + wave) * (unsigned)fa.KS * 1024u;   // this wave's subtile in the W planes
#define FE_LOAD_W(BF, s_)                                                                                        \
  _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                  \
      BF[q] = __builtin_bit_cast(fbf16x8, __builtin_amdgcn_raw_buffer_load_b128(                                 \
          wrsrc, voff, (int)(wb + q * plane_b + (unsigned)min((s_), fa.KS - 1) * 1024u), 0));
  fbf16x8 bf0[3], bf1[3];
  FE_LOAD_W(bf0, 0)
  const float *xl_f = reinterpret_cast<const float *>(s_xl);
  const float *xlg_f = reinterpret_cast<const float *>(a.x_l);
  const float *xr_f = reinterpret_cast<const float *>(a.x_r);

#pragma unroll 1
  for (int eb = 0; eb < ne; eb += FE_EB) {
    ff32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float xr0[16], xr1[16];      // the logit phase's x_r values of tile 0 / tile 1
#define FE_LOAD_XR(XR, m_)                                                                                       \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                               \
    const int t = min(eb + (m_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, ne - 1);                                  \
    XR[r] = xr_f[((size_t)(nb + s_tab[t].z) * a.ldr4 + hoff) * 4 + ccol];                                       \
  }
#pragma unroll 1
    for (int kh = 0; kh < nhalf; ++kh) {
      // B0: registers -> planes of this k half (rows beyond the graph and columns beyond K are zero)
      __syncthreads();      // phase A's tables are visible / every wave is done reading the previous planes
      if (eb > 0 && kh == 0) FE_LOAD_BLOCK(eb, FE_EID_LDS)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + FE_THREADS * u;
        const int row = i >> 4, c4 = i & 15;
        float4 v = kh == 0 ? av[u][0] : av[u][1];
        if (eb + row >= ne || kh * FE_KH + c4 * 4 >= fa.K) v = make_float4(0.f, 0.f, 0.f, 0.f);
        fbf16x4 p0, p1, p2;
        __bf16 t0, t1, t2;
        fe_split3(v.x, t0, t1, t2); p0[0] = t0; p1[0] = t1; p2[0] = t2;
        fe_split3(v.y, t0, t1, t2); p0[1] = t0; p1[1] = t1; p2[1] = t2;
        fe_split3(v.z, t0, t1, t2); p0[2] = t0; p1[2] = t1; p2[2] = t2;
        fe_split3(v.w, t0, t1, t2); p0[3] = t0; p1[3] = t1; p2[3] = t2;
        *reinterpret_cast<fbf16x4 *>(&sA[0][row][c4 * 4]) = p0;
        *reinterpret_cast<fbf16x4 *>(&sA[1][row][c4 * 4]) = p1;
        *reinterpret_cast<fbf16x4 *>(&sA[2][row][c4 * 4]) = p2;
      }
      __syncthreads();
      if (kh == nhalf - 1) FE_LOAD_XR(xr0, 0)   // tile 0's: requested now, used after this half's MFMAs
      // B1: up to four 16-deep k-steps of this half, W fragments one step ahead (the next block's first at the very end)
      const int k0 = kh * (FE_KH / 16), ksteps = min(FE_KH / 16, fa.KS - k0);
#define FE_KSTEP(BF, ks_)                                                                                        \
  {                                                                                                              \
    fbf16x8 af0[3], af1[3];   /* tile 1's fragments are requested while tile 0's MFMAs run */                   \
    _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                \
        af0[q] = *reinterpret_cast<const fbf16x8 *>(&sA[q][fr][(ks_) * 16 + fk]);                                \
    _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                \
        af1[q] = *reinterpret_cast<const fbf16x8 *>(&sA[q][32 + fr][(ks_) * 16 + fk]);                           \
    FE_MMA(0, af0, BF)                                                                                           \
    FE_MMA(1, af1, BF)                                                                                           \
  }
#ifdef FE_NO_MFMA
#undef FE_KSTEP
#define FE_KSTEP(BF, ks_) { asm volatile("" ::"v"(__builtin_bit_cast(fe_i32x4, BF[0])), "v"(__builtin_bit_cast(fe_i32x4, BF[1])), "v"(__builtin_bit_cast(fe_i32x4, BF[2]))); }
#endif
#define FE_MMA(m_, AF, BF)                                                                                       \
  {                                                                                                              \
    ff32x16 c = acc[m_];                                                                                         \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[2], c, 0, 0, 0);                                       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[2], BF[0], c, 0, 0, 0);                                       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[1], BF[1], c, 0, 0, 0);                                       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[1], c, 0, 0, 0);                                       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[1], BF[0], c, 0, 0, 0);                                       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[0], c, 0, 0, 0);                                       \
    acc[m_] = c;                                                                                                 \
  }
      int ks = 0;
#pragma unroll 1
      for (; ks + 2 <= ksteps; ks += 2) {
        const int g1 = k0 + ks + 1, g2 = k0 + ks + 2;
        FE_LOAD_W(bf1, g1)
        FE_KSTEP(bf0, ks)
        FE_LOAD_W(bf0, g2 >= fa.KS ? 0 : g2)
        FE_KSTEP(bf1, ks + 1)
      }
      if (ks < ksteps) {      // odd tail: only in the last half of K
        FE_LOAD_W(bf1, 0)
        FE_KSTEP(bf0, ks)
#pragma unroll
        for (int q = 0; q < 3; ++q) bf0[q] = bf1[q];
      }
    }
    // B2: logits.  Accumulator layout: lane -> column fr, register r -> slot (r & 3) + 8 (r >> 2) + 4 h of tile m;
    // tile 1's x_r values are requested before tile 0 is worked on
    FE_LOAD_XR(xr1, 1)
#define FE_LOGITS(XR, m_)                                                                                        \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                               \
    const int t = eb + (m_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;                                               \
    const int4 rec = s_tab[min(t, ne - 1)];                                                                      \
    const int jl = rec.x;                                                                                        \
    float xl;                                                                                                    \
    if (jl < rows) xl = xl_f[jl * (FE_RQ * 4) + ccol];                                                           \
    else xl = xlg_f[((size_t)(nb + jl) * a.ldl4 + hoff) * 4 + ccol];                                             \
    float s = (XR[r] + xl) + acc[m_][r];                                                                         \
    if (MASKED) s *= __int_as_float(rec.w);                                                                      \
    s = leaky(s, slope);                                                                                         \
    if (MASKED) s *= __int_as_float(rec.w);                                                                      \
    const float part = group_sum<32>(s * attv);                                                                  \
    if (fr == 0 && t < ne) s_part[t * FE_WAVES + wave] = part;                                                   \
  }
#ifndef FE_NO_LOGITS
    FE_LOGITS(xr0, 0)
    FE_LOGITS(xr1, 1)
#else
    if (fr == 0 && eb + 4 * h < ne) s_part[(eb + 4 * h) * FE_WAVES + wave] = xr0[0] + xr1[0] + acc[0][0] + acc[1][0];
#endif
#undef FE_LOGITS
#undef FE_LOAD_XR
  }
#undef FE_KSTEP
#undef FE_MMA
#undef FE_LOAD_W
#undef FE_LOAD_BLOCK
  __syncthreads();
  // B3: a head's four subtile partials, in subtile order
  if (tid < ne * FE_HS) {
    const int t = tid >> 1, hd2 = tid & 1;
    const float *p = s_part + t * FE_WAVES + hd2 * 4;
    s_lg[t * FE_HS + hd2] = ((p[0] + p[1]) + p[2]) + p[3];
  }
  __syncthreads();

  // ---- phase C (isg_mp_graph.hip, HS = 2, one pass: lane group of 32 = head, lane l owns float4 l) ---------------------
  const int grp = lane >> 5, l = lane & 31;
  const int off = grp * (FE_C / 4) + l;
  const int hd = hg * FE_HS + grp;
#pragma unroll 1
  for (int k = wave; k < n; k += FE_WAVES) {
    const int rb = fe_uni(s_rowptr[k]), re = min(fe_uni(s_rowptr[k + 1]), ne);
    float mx = -INFINITY;
#pragma unroll 1
    for (int t = rb; t < re; ++t) mx = fmaxf(mx, s_lg[t * FE_HS + grp]);
    float den = 0.f;
#pragma unroll 1
    for (int t = rb; t < re; ++t) den += __builtin_amdgcn_exp2f((s_lg[t * FE_HS + grp] - mx) * 1.4426950408889634f);
    const float rden = __builtin_amdgcn_rcpf(den + 1e-16f);
    float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int t = rb; t < re; ++t) {
      const int4 rec = s_tab[t];
      const float w = __builtin_amdgcn_exp2f((s_lg[t * FE_HS + grp] - mx) * 1.4426950408889634f) * rden;
      if (l == 0) a.alpha[(size_t)fe_uni(rec.y) * a.H + hd] = w;
      const float wm = MASKED ? __fmul_rn(w, __int_as_float(fe_uni(rec.w))) : w;
      const int jl = fe_uni(rec.x);
      const bool in_lds = jl < rows;
      float4 u4 = s_xl[(in_lds ? jl : 0) * FE_RQ + off];
      if (!in_lds) u4 = a.x_l[(size_t)(nb + jl) * a.ldl4 + hoff + off];
      acc4.x = __fadd_rn(acc4.x, __fmul_rn(u4.x, wm));
      acc4.y = __fadd_rn(acc4.y, __fmul_rn(u4.y, wm));
      acc4.z = __fadd_rn(acc4.z, __fmul_rn(u4.z, wm));
      acc4.w = __fadd_rn(acc4.w, __fmul_rn(u4.w, wm));
    }
    if (a.bias) {
      const float4 b = a.bias[hoff + off];
      acc4.x += b.x; acc4.y += b.y; acc4.z += b.z; acc4.w += b.w;
    }
    st_stream(a.out + (size_t)(nb + k) * R + hoff + off, acc4, nt);
  }
}

}  // namespace isg

using namespace isg;

extern "C" int isg_gatv2_mp_fused_edge_fwd(const float *x_l, const float *x_r, const float *edge_attr,
                                           const uint16_t *w_frag, const float *att, const float *bias,
                                           const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                           const float *node_mask, const float *edge_mask, float *out, float *alpha,
                                           int64_t N, int64_t E, int32_t H, int32_t C, int32_t K, float negative_slope,
                                           const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst,
                                           int64_t B, int32_t nmax_host, int32_t emax_host, int32_t ld_l, int32_t ld_r,
                                           int32_t ld_ea, void *stream) {
  if (N < 0 || E < 0 || H <= 0 || C <= 0 || K <= 0 || B < 0) return ISG_EINVAL;
  if (C != FE_C || (H & 1) || K > 128 || (K & 3) || nmax_host <= 0 || nmax_host > FE_NCAP || emax_host < 0 ||
      emax_host > FE_ECAP || B * (H / 2) >= (1ll << 31) || N >= (1ll << 31) || E >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  if (N == 0 || B == 0) return ISG_OK;
  if (!x_l || !x_r || (E > 0 && (!edge_attr || !eid || !src || !dst)) || !w_frag || !att || !rowptr || !out || !alpha ||
      !graph_ptr || !graph_eptr)
    return ISG_EINVAL;
  const int HC = H * C;
  if (ld_l < HC || ld_r < HC || (ld_l & 3) || (ld_r & 3) || ld_ea < K || (ld_ea & 3)) return ISG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x_l) | reinterpret_cast<uintptr_t>(x_r) | reinterpret_cast<uintptr_t>(edge_attr) |
       reinterpret_cast<uintptr_t>(out)) & 15)
    return ISG_EUNSUPPORTED;
  FeArgs fa;
  MpArgs &a = fa.m;
  a.x_l = reinterpret_cast<const float4 *>(x_l);
  a.x_r = reinterpret_cast<const float4 *>(x_r);
  a.e_proj = nullptr;
  a.att = reinterpret_cast<const float4 *>(att);
  a.bias = reinterpret_cast<const float4 *>(bias);
  a.rowptr = rowptr; a.eid = eid; a.src = src;
  a.node_mask = node_mask; a.edge_mask = edge_mask;
  a.out = reinterpret_cast<float4 *>(out);
  a.alpha = alpha;
  a.N = (int)N; a.C = C; a.H = H;
  a.lde4 = 0; a.ldl4 = ld_l >> 2; a.ldr4 = ld_r >> 2;
  a.slope = negative_slope;
  a.graph_ptr = graph_ptr; a.graph_eptr = graph_eptr; a.dst = dst;
  a.B = (int)B; a.f16 = 0; a.flags = 1 | 4; a.nchunks = 0; a.rowmax = nullptr;
  fa.edge_attr = edge_attr;
  fa.w_frag = reinterpret_cast<const __bf16 *>(w_frag);
  fa.K = K; fa.KS = (K + 15) / 16; fa.NT = (HC + 31) / 32; fa.ld_ea = ld_ea;
  // x_l window: what is left of ~64 KB per workgroup (2 workgroups per CU) after the static tables, at most the largest graph
  const size_t row_bytes = (size_t)FE_RQ * 16;
  a.lrows = 22;
  if (a.lrows > nmax_host) a.lrows = nmax_host;
  const size_t dyn = (size_t)a.lrows * row_bytes;
  dim3 grid((unsigned)(B * (H / 2))), block(FE_THREADS);
  hipStream_t st = as_stream(stream);
  if (node_mask || edge_mask) gatv2_mp_fused_edge_kernel<true><<<grid, block, dyn, st>>>(fa);
  else gatv2_mp_fused_edge_kernel<false><<<grid, block, dyn, st>>>(fa);
  return check_launch();
}
