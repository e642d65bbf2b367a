// One MaskingGATv2Conv per launch on graph-aligned tiles: lin_l | lin_r, lin_edge, logits, softmax, aggregation.
//
// Reference: MaskingGATv2Conv.forward after the instruction gate and the node mask, ISubGVQA/models/mgat_v2_conv.py:177-181
// (x_l = lin_l(x), x_r = lin_r(x)), :215-232 / :243-279 (message, softmax, aggregate) with lin_edge (:259-261) inside.
// isg_gatv2_tile_conv (csrc/isg_layer_tile.hip) still reads x_l / x_r [N, H*C] that a separate launch (isg_linear_f16x3, 114 us per
// layer at BASELINE configs[1], at its write floor: 337 MB) wrote just before -- 674 MB per layer of HBM traffic for two tensors
// that are consumed once, tile by tile.  Here a persistent workgroup owns one head (as there) and forms the head's x_l / x_r slices
// of a tile's <= 64 nodes itself, from the gated node rows, on the matrix cores, straight into LDS: the logit epilogue gathers
// BOTH from LDS (no per-edge row gathers from global memory at all), the aggregation reads x_l from LDS, and neither tensor ever
// exists in memory.  The node rows arrive as the scaled (hi, mid) planes their PRODUCER wrote (isg_instr_gate_planes, the previous
// layer's isg_mgat_dense_tail): the row split is done once per row, not once per (tile, head).  HBM per layer: node planes 42 MB +
// edge planes 105 MB in (both shared by a tile's four head workgroups through L2), out 168 MB + alpha 3 MB back.
//   block  = 8 waves, one workgroup per CU, persistent.  LDS: two 33 KB row slices (x_l, x_r), one 34 KB panel image (edge chunks,
//            padded rows), one 32 KB panel of node planes filled by LDS-DMA (swizzled on the source address), two sets of tables
//   wave   = node GEMM: all 64 rows x one of the head's eight 32-column tiles of [lin_l | lin_r]; edge GEMM: one 32-slot half of a
//            64-slot chunk x one of the head's four 32-channel tiles of lin_edge (transposed products, isg_mp_logits.hip);
//            softmax + aggregation: eight destination nodes per wave instruction (8 lanes x 64 bytes of a node's row)
//   W      = both weights' fragments of the wave's tiles live in its registers for the whole launch (128 VGPRs)
//   ahead  = the next tile's tables (raw values in registers), node planes (DMA) and first edge planes are requested two chunks /
//            one phase before their use; the hand-over between tiles is one barrier
// The arithmetic is isg_linear_f16x3's (row scale, planes, MFMA order, epilogue) followed by isg_gatv2_tile_conv's, operation
// for operation: out / alpha / rowmax are bit-identical to the two-launch path.  What bounds the kernel (it is the sum of its
// phases) and what was tried: DESIGN.md 9.1, profiles/r03_bg_layer_conv_ablation.md.
#include "isg_f16x3.hpp"

#include "isg_diag.hpp"

ISG_DIAG_BUFFER(g_lc_stamps)            // -DISG_DIAG builds only (tools/stamp_layer_conv.py): [workgroups * 8 waves][16] int64
#define LC_STAMP(i) ISG_DIAG_ADD(i)

namespace isg {

typedef __attribute__((address_space(3))) void lc_lds_t;
typedef __attribute__((address_space(1))) void lc_glb_t;

constexpr int LC_ROWS = 64, LC_ECAP = 256, LC_C = 128, LC_K = 128, LC_LDX = LC_C + 4, LC_LDA = LC_K + 8, LC_THREADS = 512;
constexpr int LC_OFF_XR = LC_ROWS * LC_LDX * 4;                       // x_r slice behind the x_l slice
constexpr int LC_OFF_A = 2 * LC_ROWS * LC_LDX * 4;                    // panel image: node rows, then edge chunks
constexpr int LC_OFF_TAB = LC_OFF_A + 2 * 64 * LC_LDA * 2;
constexpr int LC_OFF_RAW = LC_OFF_TAB + 2 * (LC_ECAP * 16 + LC_ECAP * 4 + 68 * 4) + LC_ECAP * 4 + 4 * 64 * 4 + 64 * 4 + 3 * LC_C * 4 +
                           4 * LC_C * 4;                              // the next tile's node rows as they come from memory (fp32)
constexpr int LC_SMEM_BYTES = LC_OFF_RAW + LC_ROWS * LC_K * 4;
static_assert(LC_OFF_RAW % 16 == 0, "16-byte pieces");
static_assert(LC_SMEM_BYTES <= 160 * 1024, "one workgroup per CU");

struct LcArgs {
  const _Float16 *xp;               // gated node rows gelu(h * instruction[batch]) as scaled (hi, mid) planes [N][2][128]
  const float *xinv;                // [N] inverse row scales (isg_instr_gate_planes / isg_mgat_dense_tail / isg_edge_planes write both)
  const _Float16 *Wn;               // [lin_l.weight; lin_r.weight] [2*H*C, 128] as fragment-major (hi, mid) planes
  const float *wn_inv, *bn;         // [2*H*C] inverse scales, biases (lin_l then lin_r)
  const _Float16 *ep;               // edge features as scaled (hi, mid) planes in CSR slot order [E][2][128] (isg_edge_planes)
  const float *ep_inv;              // [E]
  const _Float16 *We;               // lin_edge.weight [H*C, K] fragment planes
  const float *we_inv, *att;        // [H*C]
  const float *bias;                // [H*C], optional
  const int *rowptr, *eid, *src, *dst, *ntiles;
  const int4 *tile_info;
  const float *edge_mask, *node_mask;   // optional (NULL: the layer is not masked)
  float *out, *alpha;
  float *rowmax;                    // optional
  int N, E, H, KSE, NTE, ldo;
  float slope;
};

// gelu(x * instr[batch]) -> fp32 rows (optional) + scaled (hi, mid) planes + inverse row scale: 32 lanes x 4 channels per row
__global__ __launch_bounds__(256) void instr_gate_planes_kernel(const float *__restrict__ x, const float *__restrict__ instr,
                                                                const int64_t *__restrict__ batch, float *__restrict__ out,
                                                                _Float16 *__restrict__ planes, float *__restrict__ inv_out, int N) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5), c4 = threadIdx.x & 31;
  if (row >= N) return;
  const int64_t b = batch[row];
  const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)row * LC_K + c4 * 4);
  const float4 w = *reinterpret_cast<const float4 *>(instr + b * LC_K + c4 * 4);
  const isg_f32x2 g0 = gelu_exact2(isg_f32x2{v.x * w.x, v.y * w.y}), g1 = gelu_exact2(isg_f32x2{v.z * w.z, v.w * w.w});
  hf32x4 g = {g0.x, g0.y, g1.x, g1.y};
  if (out) *reinterpret_cast<hf32x4 *>(out + (int64_t)row * LC_K + c4 * 4) = g;
  const float mx = group_max<32>(fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fmaxf(fabsf(g[2]), fabsf(g[3]))));
  float sc, inv;
  h3_scale(mx, sc, inv);
  if (c4 == 0) inv_out[row] = inv;
  g *= sc;
  hf16x4 hi = {(_Float16)g[0], (_Float16)g[1], (_Float16)g[2], (_Float16)g[3]};
  hf16x4 mid = {(_Float16)(g[0] - (float)hi[0]), (_Float16)(g[1] - (float)hi[1]), (_Float16)(g[2] - (float)hi[2]),
                (_Float16)(g[3] - (float)hi[3])};
  *reinterpret_cast<hf16x4 *>(planes + (int64_t)row * 256 + c4 * 4) = hi;
  *reinterpret_cast<hf16x4 *>(planes + (int64_t)row * 256 + 128 + c4 * 4) = mid;
}

// The masked layer's node gate from the layer input's PLANES (masking.py:137, 151-155):
//   gate_n = gelu( < gelu(node_nn(x))_n , q[r(n)] > / sqrt(C) ),   r(n) = batch[n] or batch[batch[n]] (quirk Q3)
// 64 rows per workgroup (no graph alignment needed), 4 waves: the planes -- the same ones isg_gatv2_layer_conv reads -- go to LDS as
// they are, node_nn is one transposed product per wave (32 columns x 64 rows, isg_linear_f16x3's arithmetic), its GELU'd result
// stays in LDS as fp32 rows and is reduced against q exactly as node_gate_kernel does it (16 lanes per node, two float4 per lane,
// the same butterfly).  The [N, 128] intermediate is neither written nor read back, and the fp32 copy of the gated rows that the
// un-fused node_nn needed is not written by the previous layer's tail either.
constexpr int NG_ROWS = 64, NG_C = 128, NG_LDA = NG_C + 8, NG_LDC = NG_C + 4;
__global__ __launch_bounds__(256, 3) void node_gate_planes_kernel(const _Float16 *__restrict__ xp, const float *__restrict__ xinv,
                                                                  const _Float16 *__restrict__ wf, const float *__restrict__ w_inv,
                                                                  const float *__restrict__ bias, const float *__restrict__ q,
                                                                  const long long *__restrict__ batch, int dbl,
                                                                  float *__restrict__ gate, int N, float denom) {
  __shared__ __attribute__((aligned(16))) unsigned char ng_smem[2 * NG_ROWS * NG_LDA * 2];
  __shared__ float s_inv[NG_ROWS];
  __shared__ int s_b[NG_ROWS];
  typedef _Float16 (*BufP)[NG_ROWS][NG_LDA];
  BufP sA = reinterpret_cast<BufP>(ng_smem);
  float(*sC)[NG_LDC] = reinterpret_cast<float(*)[NG_LDC]>(ng_smem);       // aliases the planes once every wave is done with them
  static_assert(NG_ROWS * NG_LDC * 4 <= 2 * NG_ROWS * NG_LDA * 2, "xn aliases the plane image");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, hh = lane >> 5, fk = hh * 8;
  const int r0 = blockIdx.x * NG_ROWS, nrows = min(NG_ROWS, N - r0);
  // the wave's 32-column tile of node_nn.0's fragments: requested first
  constexpr unsigned plane = (unsigned)(NG_C / 32) * 8u * 1024u;
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(wf), 0, (int)(2u * plane), 0x00020000);
  hf16x8 wq[8][2];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      wq[ks][p] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                 wr, lane * 16, (int)(((unsigned)wave * 8u + (unsigned)ks) * 1024u + p * plane), 0));
  {
    const int srow = tid >> 5, sc4 = tid & 31;       // 32 lanes x 16 bytes = one row's 512 bytes of (hi, mid) planes
    hf32x4 pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gr = min(r0 + srow + 8 * u, N - 1);
      pv[u] = *reinterpret_cast<const hf32x4 *>(xp + (int64_t)gr * 256 + sc4 * 8);
    }
    if (tid < NG_ROWS) {
      const int gr = min(r0 + tid, N - 1);
      s_inv[tid] = xinv[gr];
      long long b = batch[gr];
      if (dbl) b = batch[min(b, (long long)N - 1)];          // batch[batch[n]] (quirk Q3)
      s_b[tid] = (int)b;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<hf32x4 *>(&sA[sc4 >> 4][srow + 8 * u][(sc4 & 15) * 8]) = pv[u];
  }
  hf32x4 wi4[4], bv4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    wi4[g] = *reinterpret_cast<const hf32x4 *>(w_inv + wave * 32 + 8 * g + 4 * hh);
    bv4[g] = *reinterpret_cast<const hf32x4 *>(bias + wave * 32 + 8 * g + 4 * hh);
  }
  __syncthreads();
  hf32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  {
    hf16x8 af[2][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[i][p] = *reinterpret_cast<const hf16x8 *>(&sA[p][i * 32 + fr][ks * 16 + fk]);
      // transposed (W fragment = A operand): a lane holds ONE row and four runs of four columns
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][1], af[i][0], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][0], acc[i], 0, 0, 0);
    }
  }
  __syncthreads();          // every wave is done with the planes: xn may overwrite them
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + fr;
    const float si = s_inv[row];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const isg_f32x2 va = gelu_exact2(isg_f32x2{(acc[i][4 * g] * si) * wi4[g][0] + bv4[g][0], (acc[i][4 * g + 1] * si) * wi4[g][1] + bv4[g][1]});
      const isg_f32x2 vb = gelu_exact2(isg_f32x2{(acc[i][4 * g + 2] * si) * wi4[g][2] + bv4[g][2], (acc[i][4 * g + 3] * si) * wi4[g][3] + bv4[g][3]});
      *reinterpret_cast<hf32x4 *>(&sC[row][wave * 32 + 8 * g + 4 * hh]) = hf32x4{va.x, va.y, vb.x, vb.y};
    }
  }
  __syncthreads();
  // node_gate_kernel's reduction: 16 lanes per node, lane l covers float4 columns l and l + 16
  const int grp = lane >> 4, l = lane & 15;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int n = wave * 16 + it * 4 + grp;
    float part = 0.f;
    if (n < nrows) {
      const float *qr = q + (int64_t)s_b[n] * NG_C;
      part += dot4_rn(*reinterpret_cast<const float4 *>(&sC[n][4 * l]), *reinterpret_cast<const float4 *>(qr + 4 * l));
      part += dot4_rn(*reinterpret_cast<const float4 *>(&sC[n][4 * (l + 16)]), *reinterpret_cast<const float4 *>(qr + 4 * (l + 16)));
    }
    const float dot = group_sum<16>(part);
    if (n < nrows && l == 0) gate[r0 + n] = gelu_libm(dot / denom);
  }
}

// KSE_T: 16-column steps of the edge product when known at compile time (8 = the 128 edge features of the model: no branch between
// the MFMAs), 0 = read it from the arguments
// SL01: 0 <= negative_slope <= 1 (the reference's 0.2): leaky_relu(z) = max(z, slope z), one instruction less per value
template <bool MASKED, int KSE_T, bool SL01>
__global__ __launch_bounds__(LC_THREADS, 2) void gatv2_layer_conv_kernel(LcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lc_smem[];
  typedef float (*BufX)[LC_LDX];
  typedef _Float16 (*BufP)[64][LC_LDA];
  BufX sXl = reinterpret_cast<BufX>(lc_smem);
  BufX sXr = reinterpret_cast<BufX>(lc_smem + LC_OFF_XR);
  BufP sA = reinterpret_cast<BufP>(lc_smem + LC_OFF_A);
  // per-tile tables, TWO sets: the next tile's are written while the slowest waves still aggregate the current one
  int4 *s_tab0 = reinterpret_cast<int4 *>(lc_smem + LC_OFF_TAB);       // [2][256] {src - r0, eid, dst - r0, mask bits}
  float *s_einv0 = reinterpret_cast<float *>(s_tab0 + 2 * LC_ECAP);     // [2][256] inverse scales of the slots' edge planes
  int *s_rp0 = reinterpret_cast<int *>(s_einv0 + 2 * LC_ECAP);          // [2][68] row pointers relative to the tile's first slot
  float *s_lg = reinterpret_cast<float *>(s_rp0 + 2 * 68);
  float *s_part = s_lg + LC_ECAP;             // [4 tile-waves][64 slots]
  float *s_inv = s_part + 4 * 64;             // inverse row scales of the node planes
  float *s_att = s_inv + 64, *s_weinv = s_att + LC_C;
  float *s_bias = s_weinv + LC_C, *s_bn = s_bias + LC_C, *s_wninv = s_bn + 2 * LC_C;             // [x_l 128 | x_r 128] biases / inverse scales of the head

  const int bid = blockIdx.x;
  const int per_xcd = gridDim.x >> 3, jx = bid >> 3;
  const int hd = jx % a.H;
  const int ngrp = gridDim.x / a.H;
  int t = (bid & 7) * (per_xcd / a.H) + jx / a.H;
  const int T = *a.ntiles;
  if (t >= T) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, tw = wave & 3;      // edge GEMM: 32-slot half, 32-channel tile
  ISG_DIAG_BEGIN()
  const int fr = lane & 31, hh = lane >> 5, fk = hh * 8;
  const int hoff = hd * LC_C;
  const int srow = tid >> 5, scol = tid & 31, sc4 = scol;    // staging map: 32 lanes per 512-byte row, rows srow + 16 u

  // ---- resident W fragments: the wave's tile of lin_edge and its tile of [lin_l | lin_r] ---------------------------------------
  const int KSE = KSE_T ? KSE_T : a.KSE;
  const unsigned plane_e = (unsigned)a.NTE * (unsigned)KSE * 1024u;
  const __amdgpu_buffer_rsrc_t wrs_e =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.We), 0, (int)(2u * plane_e), 0x00020000);
  const unsigned wb_e = (unsigned)(hd * (LC_C / 32) + tw) * (unsigned)KSE * 1024u;
  const int NTN = 2 * a.H * (LC_C / 32);
  const unsigned plane_n = (unsigned)NTN * 8u * 1024u;
  const __amdgpu_buffer_rsrc_t wrs_n =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.Wn), 0, (int)(2u * plane_n), 0x00020000);
  const int ct = wave;                             // node GEMM: column tile 0..3 = x_l channels, 4..7 = x_r channels of the head
  const unsigned wb_n = (unsigned)((ct >> 2) * a.H * (LC_C / 32) + hd * (LC_C / 32) + (ct & 3)) * 8u * 1024u;
  hf16x8 wq_e[8][2], wq_n[8][2];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      wq_n[ks][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   wrs_n, lane * 16, (int)(wb_n + q * plane_n + (unsigned)ks * 1024u), 0));
      wq_e[ks][q] = hf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (ks < KSE)
        wq_e[ks][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                     wrs_e, lane * 16, (int)(wb_e + q * plane_e + (unsigned)ks * 1024u), 0));
    }
  if (tid < LC_C) {
    s_att[tid] = a.att[hoff + tid];
    s_weinv[tid] = a.we_inv[hoff + tid];
    s_bias[tid] = a.bias ? a.bias[hoff + tid] : 0.f;
  } else if (tid < 3 * LC_C) {       // columns 0..127: x_l channels of the head, 128..255: x_r channels
    const int c = tid - LC_C;
    const int src_col = (c >> 7) * a.H * LC_C + hoff + (c & 127);
    s_bn[c] = a.bn[src_col];
    s_wninv[c] = a.wn_inv[src_col];
  }
  const float slope = a.slope;

  hf32x4 ra[4];            // edge planes of a chunk, requested a chunk (or a tile) ahead: rows srow + 16 u, 16-byte piece sc4
  // (macros, not lambdas: register arrays captured by reference end up in scratch memory)
  // RAW values only: any arithmetic on a requested value makes the compiler wait for the load where it was issued (700 cycles each)
#define LC_REQUEST_TILE(d)                           /* d = {r0, nrows, e0, ne} */                                      \
  {                                                                                                                  \
    const int r0n = (d).x, nrn = min((d).y, LC_ROWS), e0n = (d).z, nen = min((d).w, LC_ECAP);                          \
    rp_n = 0;                                                                                                        \
    if (tid <= nrn) rp_n = a.rowptr[r0n + tid];                                                                      \
    rec_n = make_int4(0, 0, 0, __float_as_int(1.f));                                                                 \
    einv_n = 1.f;                                                                                                    \
    if (tid < nen) {                                                                                                 \
      rec_n.x = a.src[e0n + tid];                                                                                    \
      rec_n.y = a.eid[e0n + tid];                                                                                    \
      rec_n.z = a.dst[e0n + tid];                                                                                    \
      einv_n = a.ep_inv[e0n + tid];                                                                                  \
    }                                                                                                                \
    xinv_n = 1.f;                                                                                                    \
    if (tid < nrn) xinv_n = a.xinv[r0n + tid];                                                                       \
    /* node planes: memory -> LDS directly, no registers (sixteen held across the chunk epilogue is where the allocator */ \
    /* ran out and evicted W fragments).  A wave instruction lands 1 KB = its two rows (srow, srow + 1), 16 bytes per   */ \
    /* lane, rows 512 bytes apart: the 16-byte pieces of a 256-byte plane row sit at position p ^ (row & 15), applied   */ \
    /* HERE, on the source address, so that the node GEMM's B fragments (32 rows x one piece) spread over the banks     */ \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                  \
      const int lrow = srow + 16 * u;                                                                                \
      const int row = min(r0n + min(lrow, max(nrn - 1, 0)), a.N - 1);                                                \
      const int piece = (sc4 & 15) ^ (lrow & 15);                                                                    \
      __builtin_amdgcn_global_load_lds((lc_glb_t *)(a.xp + (int64_t)row * 256 + (sc4 >> 4) * 128 + piece * 8),      \
                                       (lc_lds_t *)(lc_smem + LC_OFF_RAW + (2 * wave + 16 * u) * 512), 16, 0, 0);    \
    }                                                                                                                \
  }
  // second stage, a phase later (the ids have arrived): the masks behind them
#define LC_REQUEST_MASKS(d)                                                                                          \
  {                                                                                                                  \
    if (MASKED && tid < min((d).w, LC_ECAP))                                                                         \
      rec_n.w = __float_as_int(a.edge_mask ? a.edge_mask[rec_n.y] : a.node_mask[rec_n.x] * a.node_mask[rec_n.z]);      \
  }
  // the tile's first chunk of edge planes: requested AFTER the aggregation (16 registers that phase needs), still a node GEMM
  // ahead of its use
#define LC_REQUEST_PLANES(d)                                                                                         \
  {                                                                                                                  \
    const int e0n = (d).z, nen = min((d).w, LC_ECAP);                                                                \
    if (nen > 0) {                                                                                                   \
      _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                                  \
        ra[u] = *reinterpret_cast<const hf32x4 *>(a.ep + (int64_t)(e0n + min(srow + 16 * u, nen - 1)) * 256 + sc4 * 8); \
    }                                                                                                                \
  }
  // records, scales and row pointers into table set `b`; the node planes' scales
#define LC_STORE_TILE(d, b)                                                                                          \
  {                                                                                                                  \
    const int r0n = (d).x, nrn = min((d).y, LC_ROWS), e0n = (d).z;                                                   \
    if (tid <= nrn) s_rp0[(b) * 68 + tid] = rp_n - e0n;                                                              \
    if (tid < LC_ECAP) {                              /* a source outside its tile is clamped into it */             \
      rec_n.x = min(max(rec_n.x - r0n, 0), max(nrn - 1, 0));                                                         \
      rec_n.z = min(max(rec_n.z - r0n, 0), max(nrn - 1, 0));                                                         \
      s_tab0[(b) * LC_ECAP + tid] = rec_n;                                                                           \
      s_einv0[(b) * LC_ECAP + tid] = einv_n;                                                                         \
    }                                                                                                                \
    if (tid < LC_ROWS) s_inv[tid] = xinv_n;                                                                          \
  }
  // the node planes this wave asked for have landed (a builtin, not asm: the compiler must SEE the LDS-DMA retired or it drains
  // every later request early); behind the NEXT barrier every wave's have, and the next tile's node GEMM may read them
#define LC_PLANES_LANDED() ISG_WAIT(0x0F70);      /* vmcnt(0) */

  int4 desc = a.tile_info[t];
  {
    int4 rec_n;
    int rp_n;
    float einv_n, xinv_n;
    LC_REQUEST_TILE(desc)
    LC_REQUEST_MASKS(desc)
    LC_REQUEST_PLANES(desc)
    LC_STORE_TILE(desc, 0)
    LC_PLANES_LANDED()
  }
  __syncthreads();
  LC_STAMP(0)              // first tile's inputs (exposed once per workgroup)
  int cur = 0;             // table set of the tile in hand

#pragma unroll 1
  while (true) {
    const int r0 = desc.x, nrows = min(desc.y, LC_ROWS), e0 = desc.z, ne = min(desc.w, LC_ECAP);
    const int t_next = t + ngrp;
    const bool has_next = t_next < T;
    int4 dn = a.tile_info[min(t_next, T - 1)];       // the next tile's descriptor: on its way under the node GEMM, read after
                                                     // it (unconditional: a conditional load is waited for where it is issued)
    const int4 *s_tab = s_tab0 + cur * LC_ECAP;
    const float *s_einv = s_einv0 + cur * LC_ECAP;
    const int *s_rp = s_rp0 + cur * 68;

    // ---- node GEMM: [lin_l | lin_r]_head . x^T, this wave's 32 channels x all 64 nodes.  TRANSPOSED like the edge product (W
    // fragment = A operand, node panel = B operand): a lane then holds ONE node and 16 channels of it in four runs of four, which
    // go to the row-major LDS slices as 16-byte stores (with the panel as the A operand a lane holds one channel of 16 nodes:
    // 32 four-byte stores and 32 scale reads per lane, 2.9 k cycles per tile: profiles/r03_ah_*)
    {
      hf32x16 accn[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accn[i][r] = 0.f;
      hf16x8 an[2][2];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            an[i][q] = *reinterpret_cast<const hf16x8 *>(lc_smem + LC_OFF_RAW + (i * 32 + fr) * 512 + q * 256 + (((ks * 2 + hh) ^ (fr & 15)) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          hf32x16 c = accn[i];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq_n[ks][1], an[i][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq_n[ks][0], an[i][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq_n[ks][0], an[i][0], c, 0, 0, 0);
          accn[i] = c;
        }
      }
      ISG_DIAG_KEEP2(accn[0][0], accn[1][15])
      LC_STAMP(8)            // node GEMM: k loop
      // THE HAND-OVER BARRIER sits here, behind the k loop (which reads only the planes, complete since the barrier that ended
      // the previous tile's chunks, and this wave's registers): a wave that is done aggregating starts the next tile's products
      // while the slowest waves still aggregate -- matrix work under the vector phase's imbalance.  Behind it: every wave is done
      // with the previous tile's slices and tables; the next tile's tables and s_inv are complete.
      __syncthreads();
      LC_STAMP(7)            // hand-over barrier
      float(*dstx)[LC_LDX] = ct < 4 ? sXl : sXr;
      int frl = fr, hhl = hh;          // laundered: the addresses below, hoisted out of the tile loop, were spilled and came back
      asm volatile("" : "+v"(frl), "+v"(hhl));       // from scratch one dependent load at a time (700 cycles per tile)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc = ct * 32 + 4 * hhl + 8 * g;                // column of [x_l 128 | x_r 128]
        const float4 wi4 = *reinterpret_cast<const float4 *>(&s_wninv[cc]);
        const float4 bv4 = *reinterpret_cast<const float4 *>(&s_bn[cc]);
        const float wiv[4] = {wi4.x, wi4.y, wi4.z, wi4.w}, bvv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float si = s_inv[i * 32 + frl];
          hf32x4 o;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) o[jj] = (accn[i][g * 4 + jj] * si) * wiv[jj] + bvv[jj];     // both scales are powers of two
          *reinterpret_cast<hf32x4 *>(&dstx[i * 32 + frl][(ct & 3) * 32 + 4 * hhl + 8 * g]) = o;
        }
      }
    }
    LC_STAMP(9)              // node GEMM: epilogue
    // chunk 0's edge planes (requested before the previous tile's aggregation) into the panel image, free since the last chunk's
    // barrier: ONE barrier covers the slices and the first panel
    if (ne > 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<hf32x4 *>(&sA[sc4 >> 4][srow + 16 * u][(sc4 & 15) * 8]) = ra[u];
    }
    LC_STAMP(10)             // panel staging: the wait for the planes, LDS writes
    __syncthreads();         // x_l / x_r slices and chunk 0's panel complete
    LC_STAMP(1)              // node GEMM: barrier
    // The next tile's rows, records and scales are requested in the LAST chunk, behind the last request for edge planes: any
    // earlier and the chunks' waits for their planes (counted from the youngest request) wait for these too; any later (before the
    // aggregation loop) and the compiler drains every request at the loop's entry, 800 cycles per tile.
    asm volatile("" : "+v"(dn.x), "+v"(dn.y), "+v"(dn.z), "+v"(dn.w)::"memory");
    const int4 desc_n = make_int4(has_next ? __builtin_amdgcn_readfirstlane(dn.x) : 0, has_next ? __builtin_amdgcn_readfirstlane(dn.y) : 0,
                                  has_next ? __builtin_amdgcn_readfirstlane(dn.z) : 0, has_next ? __builtin_amdgcn_readfirstlane(dn.w) : 0);
    int4 rec_n;                 // defined and consumed inside this iteration (a zero descriptor requests row 0 and nothing else)
    int rp_n;
    float einv_n, xinv_n;

    // ---- 64-slot chunks: edge planes -> panel image, transposed product, logit epilogue (isg_mp_logits.hip) --------------------
    // (Running the two 32-slot halves of a chunk one phase apart -- waves 0-3 in chunk c's product while waves 4-7 are in chunk
    // c - 1's epilogue -- did not overlap the matrix and the vector pipe: 304 vs 302 us, profiles/r03_aw_*.)
    const int nchunk = (ne + 63) >> 6;
    const int creq = max(nchunk - 2, 0);       // two chunks of lead: one chunk (4 k cycles) did not cover the requests' latency
    const int prow = half * 32 + fr;
    hf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // epilogue of chunk `ch`: e = acc * s_row * s_col (exact powers of two), z = (x_r[i] + x_l[j]) + e, mask, leaky, mask, z * att in
    // 4 chains (one group of reads at a time: all four in flight are 64 registers, which the allocator took from the resident W)
#define LC_EPILOGUE_(ch, SL01)                                                                                       \
  {                                                                                                                  \
    const int slot = min(64 * (ch) + prow, ne - 1);                                                                  \
    const float sinv = s_einv[slot];                                                                                 \
    int4 rec = s_tab[slot];                                                                                          \
    const float me = own_reg(__int_as_float(rec.w));    /* a scalar: never the high dword of the record's (z, w) pair */  \
    float part[4];                                                                                                   \
    int cb = tw * 32 + 4 * hh;       /* laundered: hoisted out of the chunk loop, the 16 addresses below cost 16 registers */ \
    asm volatile("" : "+v"(cb));                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                  \
      const int cc = cb + 8 * g;                                                                                     \
      const float4 xl4 = *reinterpret_cast<const float4 *>(&sXl[rec.x][cc]);                                         \
      const float4 xr4 = *reinterpret_cast<const float4 *>(&sXr[rec.z][cc]);                                         \
      const float4 at4 = *reinterpret_cast<const float4 *>(&s_att[cc]);                                              \
      const float4 wi4 = *reinterpret_cast<const float4 *>(&s_weinv[cc]);                                            \
      const float lv[4] = {xl4.x, xl4.y, xl4.z, xl4.w}, rv[4] = {xr4.x, xr4.y, xr4.z, xr4.w};                        \
      const float atv[4] = {at4.x, at4.y, at4.z, at4.w}, wiv[4] = {wi4.x, wi4.y, wi4.z, wi4.w};                      \
      part[g] = 0.f;                                                                                                 \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                             \
        const float e = (acc[g * 4 + jj] * sinv) * wiv[jj];                                                          \
        float z = (rv[jj] + lv[jj]) + e;                                                                             \
        if (MASKED) z *= me;                                                                                         \
        z = (SL01) ? fmaxf(z, z * slope) : (z > 0.f ? z : z * slope);   /* 0 <= slope <= 1: max(z, slope z), same bits */ \
        if (MASKED) z *= me;                                                                                         \
        part[g] = fmaf(z, atv[jj], part[g]);                                                                         \
      }                                                                                                              \
      if (g < 3) __builtin_amdgcn_sched_barrier(0);                                                                  \
    }                                                                                                                \
    const float mine = (part[0] + part[1]) + (part[2] + part[3]);                                                    \
    const float tot = mine + __shfl_xor(mine, 32);                                                                   \
    if (hh == 0) s_part[tw * 64 + prow] = tot;                                                                       \
  }
#define LC_EPILOGUE(ch) LC_EPILOGUE_(ch, SL01)
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
      if (c > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<hf32x4 *>(&sA[sc4 >> 4][srow + 16 * u][(sc4 & 15) * 8]) = ra[u];
        LC_STAMP(10)         // panel staging: the wait for the planes, LDS writes
        __syncthreads();
        LC_STAMP(2)          // staging barrier
      }
      if (c == creq) LC_REQUEST_TILE(desc_n)
      if (c + 1 < nchunk) {      // the next chunk's planes: in flight under this chunk's product and epilogue
#pragma unroll
        for (int u = 0; u < 4; ++u)
          ra[u] = *reinterpret_cast<const hf32x4 *>(a.ep + (int64_t)(e0 + min(64 * (c + 1) + srow + 16 * u, ne - 1)) * 256 + sc4 * 8);
      }
      // a chunk's upper 32 slots are often past the tile's end (its last chunk holds ne mod 64 slots): the four waves of that half
      // then skip product and epilogue, and the other four have their SIMDs to themselves
      const bool live = half == 0 || 64 * c + 32 < ne;
      if (live) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        hf16x8 af[2][2];          // [stage][plane]
#pragma unroll
        for (int q = 0; q < 2; ++q) af[0][q] = *reinterpret_cast<const hf16x8 *>(&sA[q][prow][fk]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks < KSE) {
            if (ks + 1 < 8) {
#pragma unroll
              for (int q = 0; q < 2; ++q)
                af[(ks + 1) & 1][q] = *reinterpret_cast<const hf16x8 *>(&sA[q][prow][(ks + 1) * 16 + fk]);
            }
            // transposed product: W fragment = A operand (rows = channels), edge panel = B operand (columns = edges)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq_e[ks][0], af[ks & 1][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq_e[ks][1], af[ks & 1][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq_e[ks][0], af[ks & 1][0], acc, 0, 0, 0);
          }
        }
      }
      ISG_DIAG_KEEP2(acc[0], acc[15])
      LC_STAMP(3)            // k loop
      if (MASKED && c + 1 == nchunk) LC_REQUEST_MASKS(desc_n)      // the ids they hang on were requested a k loop ago
      if (live) LC_EPILOGUE(c)
      __syncthreads();
      LC_STAMP(4)            // epilogue + barrier
      if (tid < 64 && 64 * c + tid < ne)     // the tile-waves' partials in a fixed order
        s_lg[64 * c + tid] = (s_part[tid] + s_part[64 + tid]) + (s_part[128 + tid] + s_part[192 + tid]);
    }
#undef LC_EPILOGUE
#undef LC_EPILOGUE_
    LC_PLANES_LANDED()       // requested two chunks ago
    __syncthreads();

    // ---- softmax + aggregation (isg_mp_graph.hip phase C: same operations in the same order) ----------------------------------
    // One pass per destination node, 32 lanes x 16 bytes of its output row: every lane repeats the segment's max / denominator /
    // weights (the logits are broadcast reads), so no weight table and no barrier between the softmax and the aggregation; the
    // first four in-edges (most segments) are read once and stay in registers for all three uses.
    LC_STAMP(11)             // last logit sums + barrier
    if (nchunk == 0) {          // a tile without edges: nothing hid the requests (uniform over the workgroup)
      LC_REQUEST_TILE(desc_n)
      LC_REQUEST_MASKS(desc_n)
      LC_PLANES_LANDED()
      __syncthreads();
    }
    LC_STORE_TILE(desc_n, cur ^ 1)      // panel image and scales are free since the last chunk's barrier; tables: the other set
    LC_REQUEST_PLANES(desc_n)
    LC_STAMP(5)              // the next tile's planes and tables
    // 8 lanes per node (four 16-byte pieces each, 128 contiguous bytes per instruction): a wave aggregates EIGHT nodes at a time,
    // ONE pass covers the tile.  The phase is issue-bound on its per-node bookkeeping (bounds, records, softmax, stores), which a
    // pass does once per instruction whatever the lane count per node: 16 lanes per node needed two passes, 32 lanes four.
    {
      const int q8 = lane >> 3, j8 = lane & 7;
      const int k = 8 * wave + q8;
      if (k < nrows) {
        const int rb = s_rp[k], re = min(s_rp[k + 1], ne);
        float lg4[4], e4[4];
        int4 rc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = max(min(rb + u, re - 1), 0);
          lg4[u] = s_lg[idx];
          rc4[u] = s_tab[idx];
        }
        float4 u2[2][4];          // the rows of two in-edges at a time (four would be 64 registers)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int m = 0; m < 4; ++m) u2[u][m] = *reinterpret_cast<const float4 *>(&sXl[rc4[u].x][32 * m + j8 * 4]);
        float mx = fmaxf(fmaxf(-INFINITY, lg4[0]), fmaxf(fmaxf(lg4[1], lg4[2]), lg4[3]));
#pragma unroll 1
        for (int s = rb + 4; s < re; s += 4) {
          const float v0 = s_lg[s], v1 = s_lg[min(s + 1, re - 1)], v2 = s_lg[min(s + 2, re - 1)], v3 = s_lg[min(s + 3, re - 1)];
          mx = fmaxf(fmaxf(mx, v0), fmaxf(fmaxf(v1, v2), v3));
        }
        float den = 0.f;          // in slot order, like the per-edge loop of the kernels this replaces
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          e4[u] = __builtin_amdgcn_exp2f((lg4[u] - mx) * 1.4426950408889634f);
          den += rb + u < re ? e4[u] : 0.f;
        }
#pragma unroll 1
        for (int s = rb + 4; s < re; s += 4) {
          const float v0 = s_lg[s], v1 = s_lg[min(s + 1, re - 1)], v2 = s_lg[min(s + 2, re - 1)], v3 = s_lg[min(s + 3, re - 1)];
          const float e0 = __builtin_amdgcn_exp2f((v0 - mx) * 1.4426950408889634f);
          const float e1 = __builtin_amdgcn_exp2f((v1 - mx) * 1.4426950408889634f);
          const float e2 = __builtin_amdgcn_exp2f((v2 - mx) * 1.4426950408889634f);
          const float e3 = __builtin_amdgcn_exp2f((v3 - mx) * 1.4426950408889634f);
          den += e0;
          den += s + 1 < re ? e1 : 0.f;
          den += s + 2 < re ? e2 : 0.f;
          den += s + 3 < re ? e3 : 0.f;
        }
        const float rden = __builtin_amdgcn_rcpf(den + 1e-16f);
        float4 o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) o[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = rb;
#pragma unroll 1
        while (true) {          // four in-edges per round (two and two), edge-id order, one fma per term like every kernel of the family
          {     // alpha: ONE store instruction per round, lane u of a node's eight writes in-edge u
            const float wsel = (j8 & 2) ? ((j8 & 1) ? e4[3] : e4[2]) : ((j8 & 1) ? e4[1] : e4[0]);
            const int esel = (j8 & 2) ? ((j8 & 1) ? rc4[3].y : rc4[2].y) : ((j8 & 1) ? rc4[1].y : rc4[0].y);
            if (j8 < 4 && s + j8 < re) a.alpha[(int64_t)esel * a.H + hd] = wsel * rden;
          }
#pragma unroll
          for (int hp = 0; hp < 2; ++hp) {
            if (hp == 1) {      // the second pair's rows: only when some node of the wave has a third in-edge in this round
              if (!__any(s + 2 < re)) break;
#pragma unroll
              for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) u2[u][m] = *reinterpret_cast<const float4 *>(&sXl[rc4[2 + u].x][32 * m + j8 * 4]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              // slots past the segment's end carry weight 0 instead of a branch each: their rows are copies of the last slot's
              // (finite), and x + 0 * r = x exactly
              const float w = s + 2 * hp + u < re ? e4[2 * hp + u] * rden : 0.f;
              const float wm = MASKED ? mul_rn(w, __int_as_float(rc4[2 * hp + u].w)) : w;
#pragma unroll
              for (int m = 0; m < 4; ++m) {
                o[m].x = fmaf(u2[u][m].x, wm, o[m].x);
                o[m].y = fmaf(u2[u][m].y, wm, o[m].y);
                o[m].z = fmaf(u2[u][m].z, wm, o[m].z);
                o[m].w = fmaf(u2[u][m].w, wm, o[m].w);
              }
            }
          }
          s += 4;
          if (s >= re) break;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = min(s + u, re - 1);
            e4[u] = __builtin_amdgcn_exp2f((s_lg[idx] - mx) * 1.4426950408889634f);
            rc4[u] = s_tab[idx];
          }
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) u2[u][m] = *reinterpret_cast<const float4 *>(&sXl[rc4[u].x][32 * m + j8 * 4]);
        }
        float rmx = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (a.bias) {
            const float4 b4 = *reinterpret_cast<const float4 *>(&s_bias[32 * m + j8 * 4]);
            o[m].x += b4.x; o[m].y += b4.y; o[m].z += b4.z; o[m].w += b4.w;
          }
          hf32x4 o4 = {o[m].x, o[m].y, o[m].z, o[m].w};
          __builtin_nontemporal_store(o4, reinterpret_cast<hf32x4 *>(a.out + (int64_t)(r0 + k) * a.ldo + hoff + 32 * m + j8 * 4));
          rmx = fmaxf(rmx, fmaxf(fmaxf(fabsf(o[m].x), fabsf(o[m].y)), fmaxf(fabsf(o[m].z), fabsf(o[m].w))));
        }
        if (a.rowmax) {
          rmx = group_max<8>(rmx);
          if (j8 == 0) a.rowmax[(int64_t)(r0 + k) * a.H + hd] = rmx;
        }
      }
    }
    LC_STAMP(6)              // softmax + aggregation, stores
    if (!has_next) break;
    desc = desc_n;           // (no barrier here: it waits behind the next tile's node products)
    t = t_next;
    cur ^= 1;
  }
#undef LC_REQUEST_TILE
#undef LC_REQUEST_PLANES
#undef LC_REQUEST_MASKS
#undef LC_STORE_TILE
#undef LC_PLANES_LANDED
  ISG_DIAG_DUMP(g_lc_stamps, bid * 8 + wave, 12, )
}

}  // namespace isg

using namespace isg;

ISG_DIAG_SETTER(isg_lc_set_stamp_buffer, g_lc_stamps)

// gelu(x * instr[batch]) (mgat_v2_conv.py:156-157, isg_instr_gate) written as the (hi, mid) planes + inverse row scales that
// isg_gatv2_layer_conv reads, and as fp32 rows too when `out` is given (the masked layer's node gate reads those): the same row
// scale / split as isg_linear_f16x3's staging, done once per row here instead of once per (tile, head) in the layer kernel.
extern "C" int isg_instr_gate_planes(const float *x, const float *instr, const int64_t *batch, float *out, uint16_t *planes,
                                     float *inv_scale, int64_t N, int32_t C, void *stream) {
  if (N < 0 || C <= 0) return ISG_EINVAL;
  if (C != LC_K || (reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(instr) & 15) != 0 ||
      (out && (reinterpret_cast<uintptr_t>(out) & 15) != 0) || (reinterpret_cast<uintptr_t>(planes) & 15) != 0 || N >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  if (N == 0) return ISG_OK;
  if (!x || !instr || !batch || !planes || !inv_scale) return ISG_EINVAL;
  instr_gate_planes_kernel<<<(unsigned)((N + 7) / 8), 256, 0, as_stream(stream)>>>(x, instr, batch, out, reinterpret_cast<_Float16 *>(planes),
                                                                                   inv_scale, (int)N);
  return check_launch();
}

// The masked layer's node gate (masking.py:137, 151-155) from the layer input's planes, node_nn inside: see node_gate_planes_kernel.
// x_planes / x_inv_scale as isg_gatv2_layer_conv's; w_frag / w_inv_scale = isg_split_f16x2_frag of node_nn.0.weight [128,128], b its
// bias; q fp32 [Bq,128] = ques_nn(u); gate fp32 [N] = isg_node_gate(gelu(node_nn(x)), q, ...) with the Linear on the fp16 three-
// product form.  ISG_EUNSUPPORTED unless C == 128.
extern "C" int isg_node_gate_planes(const uint16_t *x_planes, const float *x_inv_scale, const uint16_t *w_frag, const float *w_inv_scale,
                                    const float *b, const float *q, const int64_t *batch, int32_t double_index, float *gate,
                                    int64_t N, int32_t C, void *stream) {
  if (N < 0 || C <= 0) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (C != NG_C || mis(x_planes) || mis(w_inv_scale) || mis(b) || mis(q) || N >= (1ll << 31)) return ISG_EUNSUPPORTED;
  if (N == 0) return ISG_OK;
  if (!x_planes || !x_inv_scale || !w_frag || !w_inv_scale || !b || !q || !batch || !gate) return ISG_EINVAL;
  node_gate_planes_kernel<<<(unsigned)((N + NG_ROWS - 1) / NG_ROWS), 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const _Float16 *>(x_planes), x_inv_scale, reinterpret_cast<const _Float16 *>(w_frag), w_inv_scale, b, q,
      reinterpret_cast<const long long *>(batch), double_index, gate, (int)N, sqrtf((float)C));   // torch.sqrt(torch.tensor(C)): fp32
  return check_launch();
}

// lin_l | lin_r + MaskingGATv2Conv.message + aggregate (lin_edge inside) as one persistent launch: see the file header.
// x_planes [N][2][128] fp16 + x_inv_scale [N] = the gated layer input gelu(h * instruction[batch]) as scaled (hi, mid) planes
// (isg_instr_gate_planes, isg_mgat_dense_tail's xp_out, or isg_edge_planes with eid = NULL on fp32 rows); wn_frag / wn_inv_scale =
// isg_split_f16x2_frag of cat(lin_l.weight, lin_r.weight) [2*H*C, 128], bn fp32 [2*H*C] their biases; the rest as isg_gatv2_tile_conv.
extern "C" int isg_gatv2_layer_conv(const uint16_t *x_planes, const float *x_inv_scale, const uint16_t *wn_frag, const float *wn_inv_scale,
                                    const float *bn, const uint16_t *edge_planes, const float *edge_inv_scale, const uint16_t *we_frag,
                                    const float *we_inv_scale, const float *att, const float *bias, const int32_t *rowptr,
                                    const int32_t *eid, const int32_t *src, const int32_t *dst, const int32_t *tile_info,
                                    const int32_t *ntiles, int64_t max_tiles, const float *node_mask, const float *edge_mask,
                                    float *out, int32_t ldo, float *alpha, float *rowmax, int64_t N, int64_t E, int32_t H,
                                    int32_t C, int32_t K_in, int32_t K_edge, float negative_slope, void *stream) {
  if (N < 0 || E < 0 || H <= 0 || C <= 0 || K_in <= 0 || K_edge <= 0 || max_tiles < 0 || ldo < H * C) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (C != LC_C || K_in != LC_K || K_edge > LC_K || (K_edge & 3) != 0 || (ldo & 3) != 0 || H > 16 || mis(x_planes) ||
      mis(edge_planes) || mis(out) || (bias && mis(bias)) || mis(tile_info) || N >= (1ll << 31) || E >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  if (N == 0 || max_tiles == 0) return ISG_OK;
  if (!x_planes || !x_inv_scale || !wn_frag || !wn_inv_scale || !bn ||
      (E > 0 && (!edge_planes || !edge_inv_scale || !eid || !src || !dst || !alpha)) || !we_frag || !we_inv_scale || !att || !rowptr ||
      !tile_info || !ntiles || !out)
    return ISG_EINVAL;
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  LcArgs a = {
      .xp = reinterpret_cast<const _Float16 *>(x_planes), .xinv = x_inv_scale, .Wn = reinterpret_cast<const _Float16 *>(wn_frag),
      .wn_inv = wn_inv_scale, .bn = bn, .ep = reinterpret_cast<const _Float16 *>(edge_planes), .ep_inv = edge_inv_scale,
      .We = reinterpret_cast<const _Float16 *>(we_frag), .we_inv = we_inv_scale, .att = att, .bias = bias, .rowptr = rowptr,
      .eid = eid, .src = src, .dst = dst, .ntiles = ntiles, .tile_info = reinterpret_cast<const int4 *>(tile_info),
      .edge_mask = edge_mask, .node_mask = node_mask, .out = out, .alpha = alpha, .rowmax = rowmax, .N = (int)N, .E = (int)E,
      .H = H, .KSE = (K_edge + 15) / 16, .NTE = H * C / 32, .ldo = ldo, .slope = negative_slope};
  if (!a.xp || !a.xinv || !a.Wn || !a.wn_inv || !a.bn || (a.E > 0 && (!a.ep || !a.ep_inv || !a.eid || !a.src || !a.dst || !a.alpha)) ||
      !a.We || !a.we_inv || !a.att || !a.rowptr || !a.tile_info || !a.ntiles || !a.out)
    return ISG_EINVAL;                         // the struct the kernel dereferences, not the parameters it was filled from
  const int cus = device_cus();
  int gpx = (cus / 8) / H;                                // groups per XCD: one workgroup per CU
  if (gpx < 1) gpx = 1;
  const long long need = (max_tiles + 7) / 8;
  if (gpx > need) gpx = (int)need;
  const unsigned grid = 8u * (unsigned)H * (unsigned)gpx;
  hipStream_t st = as_stream(stream);
#define LC_LAUNCH(M, KT, SL)                                                                                         \
  {                                                                                                                  \
    if (!dyn_lds_ok<&gatv2_layer_conv_kernel<M, KT, SL>>(LC_SMEM_BYTES)) return ISG_EUNSUPPORTED;                    \
    gatv2_layer_conv_kernel<M, KT, SL><<<grid, LC_THREADS, LC_SMEM_BYTES, st>>>(a);                                  \
  }
  const bool masked = node_mask || edge_mask;
  const bool sl01 = negative_slope >= 0.f && negative_slope <= 1.f;
  if (a.KSE == 8 && sl01) {
    if (masked) LC_LAUNCH(true, 8, true) else LC_LAUNCH(false, 8, true)
  } else if (sl01) {
    if (masked) LC_LAUNCH(true, 0, true) else LC_LAUNCH(false, 0, true)
  } else {
    if (masked) LC_LAUNCH(true, 0, false) else LC_LAUNCH(false, 0, false)
  }
#undef LC_LAUNCH
  return check_launch();
}

// =====================================================================================================================
// The read-out on the same tiles: GlobalAttention.forward (ISubGVQA/models/att_pooling.py:57-77) as one launch.
//     xn = node_nn(x) = Linear(GELU(Linear(x)));  xn *= node_mask;  gate = softmax_g(<xn_n, q_g> / sqrt(C)) (+1e-16);
//     out_g = sum_n gate_n xn_n
// Un-fused: two [N,128] x [128,128] Linears (bandwidth-bound passes over 42 MB each way) + isg_global_attn_pool = 71 us at BASELINE
// configs[1].  Here a tile's rows are read once: planes -> GEMM1 -> GELU -> planes (row scale from the bound amax_i * max_j
// ||W1_j||_1 + max |b1|, as in isg_mgat_dense_tail) -> GEMM2 -> xn (masked) in LDS -> isg_global_attn_pool's arithmetic per graph.
// Only `gate` [N] and `out` [B,128] leave the kernel.
// =====================================================================================================================
namespace isg {

constexpr int RO_ROWS = 64, RO_C = 128, RO_LDA = RO_C + 8, RO_LDC = RO_C + 4, RO_GPC = 128;
constexpr int RO_SMEM_BYTES = 2 * (2 * RO_ROWS * RO_LDA * 2) + (6 * 64 + RO_GPC + 4) * 4;      // two plane images + tables: 71,696
static_assert(RO_ROWS * RO_LDC * 4 <= 2 * RO_ROWS * RO_LDA * 2, "xn aliases the first plane image");

struct RoArgs {
  const float *x;                   // [N, 128], row stride ldx
  const _Float16 *w1f, *w2f;        // node_nn.0 / node_nn.2 weights [128,128] as fragment planes
  const float *w1_inv, *b1, *w2_inv, *b2;
  const float *y_bound;             // [2]: max_j sum_k |W1[j,k]|, max_j |b1[j]|
  const float *q;                   // [B, 128] = ques_nn(u)
  const float *node_mask;           // [N] or NULL
  float *out, *gate;                // [B, 128], [N]
  const int *ptr, *tile_ptr, *ntiles;
  const int4 *tile_info;
  const long long *batch;
  int N, ldx;
  float denom;
};

__global__ __launch_bounds__(256, 2) void readout_tile_kernel(RoArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ro_smem[];
  typedef _Float16 (*BufP)[RO_ROWS][RO_LDA];          // [plane][row][k]
  BufP sA = reinterpret_cast<BufP>(ro_smem);
  BufP sB = reinterpret_cast<BufP>(ro_smem + 2 * RO_ROWS * RO_LDA * 2);
  float(*sC)[RO_LDC] = reinterpret_cast<float(*)[RO_LDC]>(ro_smem);
  float *s_f = reinterpret_cast<float *>(ro_smem + 2 * (2 * RO_ROWS * RO_LDA * 2));
  float *s_inv1 = s_f, *s_scale2 = s_f + 64, *s_inv2 = s_f + 128, *s_a = s_f + 192, *s_mask = s_f + 256;
  int *s_gid = reinterpret_cast<int *>(s_f + 320), *s_gp = reinterpret_cast<int *>(s_f + 384);

  const int t = blockIdx.x;
  const int4 tinfo = a.tile_info[t];
  if (t >= *a.ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, hh = lane >> 5, fk = hh * 8;
  const int r0 = tinfo.x, nrows = min(tinfo.y, RO_ROWS);
  const int g0 = a.tile_ptr[t], ng = a.tile_ptr[t + 1] - g0;
  if (nrows <= 0) {       // graphs without nodes: scatter_add leaves their rows at zero
    for (int i = tid; i < ng * RO_C; i += 256) a.out[(int64_t)g0 * RO_C + i] = 0.f;
    return;
  }
  const int srow = tid >> 5, sc4 = tid & 31;

  // both weights' fragments of this wave's 32-column tile: requested first
  constexpr unsigned plane = (unsigned)(RO_C / 32) * 8u * 1024u;
  const __amdgpu_buffer_rsrc_t wr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.w1f), 0, (int)(2u * plane), 0x00020000);
  const __amdgpu_buffer_rsrc_t wr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.w2f), 0, (int)(2u * plane), 0x00020000);
  hf16x8 wq[8][2];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      wq[ks][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                 wr1, lane * 16, (int)(((unsigned)wave * 8u + (unsigned)ks) * 1024u + q * plane), 0));
  // the tile's rows -> row scale -> (hi, mid) planes; tables
  {
    hf32x4 xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gr = min(r0 + min(srow + 8 * u, nrows - 1), a.N - 1);
      xv[u] = *reinterpret_cast<const hf32x4 *>(a.x + (int64_t)gr * a.ldx + sc4 * 4);
    }
    if (tid < RO_ROWS) {
      const int gr = min(r0 + min(tid, nrows - 1), a.N - 1);
      s_mask[tid] = a.node_mask ? a.node_mask[gr] : 1.f;
      s_gid[tid] = (int)a.batch[gr];
    }
    if (tid <= min(ng, RO_GPC)) s_gp[tid] = a.ptr[g0 + tid] - r0;
    const float yb0 = a.y_bound[0], yb1 = a.y_bound[1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = srow + 8 * u;
      hf32x4 v = xv[u];
      if (row >= nrows) v = hf32x4{0.f, 0.f, 0.f, 0.f};
      const float mx = group_max<32>(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      float s, inv;
      h3_scale(mx, s, inv);
      if (sc4 == 0) {
        s_inv1[row] = inv;
        float s2, inv2;
        h3_scale(fmaf(mx, yb0, yb1), s2, inv2);      // |gelu(z)| <= |z| <= amax * max_j ||W1_j||_1 + max |b1|
        s_scale2[row] = s2;
        s_inv2[row] = inv2;
      }
      v *= s;
      hf16x4 hi = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
      hf16x4 mid = {(_Float16)(v[0] - (float)hi[0]), (_Float16)(v[1] - (float)hi[1]), (_Float16)(v[2] - (float)hi[2]),
                    (_Float16)(v[3] - (float)hi[3])};
      *reinterpret_cast<hf16x4 *>(&sA[0][row][sc4 * 4]) = hi;
      *reinterpret_cast<hf16x4 *>(&sA[1][row][sc4 * 4]) = mid;
    }
  }
  __syncthreads();

  // ---- GEMM1 [64 x 128] . W1^T (this wave's 32 columns), GELU, planes of the intermediate into the second image ------------
  {
    hf32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    hf16x8 af[2][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) af[i][q] = *reinterpret_cast<const hf16x8 *>(&sA[q][i * 32 + fr][ks * 16 + fk]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][1], af[i][0], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][0], acc[i], 0, 0, 0);
    }
    // the second weight's fragments take over the registers while the epilogue runs
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        wq[ks][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   wr2, lane * 16, (int)(((unsigned)wave * 8u + (unsigned)ks) * 1024u + q * plane), 0));
    // transposed products (W fragment = A operand, as in isg_mgat_dense_tail): a lane holds ONE row and four runs of four columns
    hf32x4 wi4[4], bv4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      wi4[g] = *reinterpret_cast<const hf32x4 *>(a.w1_inv + wave * 32 + 8 * g + 4 * hh);
      bv4[g] = *reinterpret_cast<const hf32x4 *>(a.b1 + wave * 32 + 8 * g + 4 * hh);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 32 + fr;
      const float si = s_inv1[row], s2 = s_scale2[row];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const isg_f32x2 va = gelu_exact2(isg_f32x2{(acc[i][4 * g] * si) * wi4[g][0] + bv4[g][0], (acc[i][4 * g + 1] * si) * wi4[g][1] + bv4[g][1]});
        const isg_f32x2 vb = gelu_exact2(isg_f32x2{(acc[i][4 * g + 2] * si) * wi4[g][2] + bv4[g][2], (acc[i][4 * g + 3] * si) * wi4[g][3] + bv4[g][3]});
        const float y0 = va.x * s2, y1 = va.y * s2, y2 = vb.x * s2, y3 = vb.y * s2;
        const hf16x4 hi = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
        const hf16x4 mid = {(_Float16)(y0 - (float)hi[0]), (_Float16)(y1 - (float)hi[1]), (_Float16)(y2 - (float)hi[2]), (_Float16)(y3 - (float)hi[3])};
        const int c0 = wave * 32 + 8 * g + 4 * hh;
        *reinterpret_cast<hf16x4 *>(&sB[0][row][c0]) = hi;
        *reinterpret_cast<hf16x4 *>(&sB[1][row][c0]) = mid;
      }
    }
  }
  __syncthreads();          // the intermediate is complete; every wave is done with the first image (xn will overwrite it)

  // ---- GEMM2 -> xn = (. + b2) * mask into LDS -------------------------------------------------------------------------------
  {
    hf32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    hf16x8 af[2][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) af[i][q] = *reinterpret_cast<const hf16x8 *>(&sB[q][i * 32 + fr][ks * 16 + fk]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][1], af[i][0], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][1], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][0], acc[i], 0, 0, 0);
    }
    hf32x4 wi4[4], bv4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      wi4[g] = *reinterpret_cast<const hf32x4 *>(a.w2_inv + wave * 32 + 8 * g + 4 * hh);
      bv4[g] = *reinterpret_cast<const hf32x4 *>(a.b2 + wave * 32 + 8 * g + 4 * hh);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 32 + fr;
      const float si = s_inv2[row], mk = s_mask[row];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        hf32x4 v;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float t = (acc[i][4 * g + jj] * si) * wi4[g][jj] + bv4[g][jj];
          v[jj] = a.node_mask ? mul_rn(t, mk) : t;      // att_pooling.py:63: x = node_nn(x) * node_mask
        }
        *reinterpret_cast<hf32x4 *>(&sC[row][wave * 32 + 8 * g + 4 * hh]) = v;
      }
    }
  }
  __syncthreads();

  // ---- isg_global_attn_pool's arithmetic on the tile's graphs (rows from LDS) --------------------------------------------------
  // logits: a half-wave per node (the 32-lane butterfly == the old 64-lane one whose upper half adds zeros)
  for (int kb = 8 * wave; kb < nrows; kb += 32) {
    float part[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(kb + 2 * u + hh, nrows - 1);
      const float4 q4 = *reinterpret_cast<const float4 *>(a.q + (int64_t)s_gid[k] * RO_C + fr * 4);
      const float4 v = *reinterpret_cast<const float4 *>(&sC[k][fr * 4]);
      part[u] = 0.f + dot4_rn(v, q4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) part[u] = group_sum<32>(part[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + 2 * u + hh;
      if (fr == 0 && k < nrows) s_a[k] = part[u] / a.denom;
    }
  }
  __syncthreads();
  // softmax over a graph's nodes (PyG form: + 1e-16 in the denominator), a wave per graph; the gate goes out
  for (int gi = wave; gi < ng; gi += 4) {
    const int nb = gi < RO_GPC ? s_gp[gi] : a.ptr[g0 + gi] - r0;
    const int n = min(gi < RO_GPC ? s_gp[gi + 1] : a.ptr[g0 + gi + 1] - r0, nrows) - nb;
    if (n <= 0) continue;
    float *sa_g = s_a + nb;
    float mx = lane < n ? sa_g[lane] : -INFINITY;
    mx = wave_max(mx);
    if (lane < n) sa_g[lane] = expf(sa_g[lane] - mx);
    __builtin_amdgcn_wave_barrier();
    float sum = 0.f;
    for (int k = 0; k < n; ++k) sum += sa_g[k];
    sum += 1e-16f;
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
      const float w = sa_g[lane] / sum;
      sa_g[lane] = w;
      a.gate[r0 + nb + lane] = w;
    }
  }
  __syncthreads();
  // pooled sum per (graph, channel), node order, unfused mul + add
  {
    const int ch = tid & (RO_C - 1);
    for (int gi = tid >> 7; gi < ng; gi += 2) {
      const int nb = gi < RO_GPC ? s_gp[gi] : a.ptr[g0 + gi] - r0;
      const int n = min(gi < RO_GPC ? s_gp[gi + 1] : a.ptr[g0 + gi + 1] - r0, nrows) - nb;
      float sum = 0.f;
#pragma unroll 4
      for (int k = 0; k < n; ++k) sum = add_rn(sum, mul_rn(s_a[nb + k], sC[nb + k][ch]));
      a.out[(int64_t)(g0 + gi) * RO_C + ch] = sum;
    }
  }
}

}  // namespace isg

// GlobalAttention.forward (att_pooling.py:57-77) on graph-aligned 64-row tiles: node_nn (Linear GELU Linear), mask, per-graph softmax
// pooling.  x fp32 [N,128]; w1 / w2 = isg_split_f16x2_frag planes of node_nn.0 / node_nn.2 weights [128,128]; y_bound fp32 [2] as in
// isg_mgat_dense_tail; q fp32 [B,128] = ques_nn(u); out fp32 [B,128], gate fp32 [N].  ISG_EUNSUPPORTED unless C == 128.
extern "C" int isg_readout_tile(const float *x, int32_t ldx, const uint16_t *w1_frag, const float *w1_inv_scale, const float *b1,
                                const float *y_bound, const uint16_t *w2_frag, const float *w2_inv_scale, const float *b2,
                                const float *q, const float *node_mask, float *out, float *gate, const int32_t *ptr,
                                const int64_t *batch, const int32_t *tile_ptr, const int32_t *tile_info, const int32_t *ntiles,
                                int64_t max_tiles, int64_t N, int32_t C, void *stream) {
  if (N < 0 || max_tiles < 0 || ldx < C || C <= 0) return ISG_EINVAL;
  if (C != isg::RO_C || (ldx & 3) != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(q) & 15) != 0 ||
      ((reinterpret_cast<uintptr_t>(w1_inv_scale) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(w2_inv_scale) |
        reinterpret_cast<uintptr_t>(b2)) & 15) != 0 ||          // the epilogues read them as 16-byte runs
      N >= (1ll << 31) || max_tiles >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  if (max_tiles == 0) return ISG_OK;
  if (!x || !w1_frag || !w1_inv_scale || !b1 || !y_bound || !w2_frag || !w2_inv_scale || !b2 || !q || !out || !gate || !ptr || !batch ||
      !tile_ptr || !tile_info || !ntiles)
    return ISG_EINVAL;
  if (!isg::dyn_lds_ok<&isg::readout_tile_kernel>(isg::RO_SMEM_BYTES)) return ISG_EUNSUPPORTED;
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  isg::RoArgs a = {
      .x = x, .w1f = reinterpret_cast<const _Float16 *>(w1_frag), .w2f = reinterpret_cast<const _Float16 *>(w2_frag),
      .w1_inv = w1_inv_scale, .b1 = b1, .w2_inv = w2_inv_scale, .b2 = b2, .y_bound = y_bound, .q = q, .node_mask = node_mask,
      .out = out, .gate = gate, .ptr = ptr, .tile_ptr = tile_ptr, .ntiles = ntiles,
      .tile_info = reinterpret_cast<const int4 *>(tile_info), .batch = reinterpret_cast<const long long *>(batch),
      .N = (int)N, .ldx = ldx, .denom = sqrtf((float)C)};     // att_pooling.py:68 divides by torch.sqrt(torch.tensor(C)): fp32 sqrt
  if (!a.x || !a.w1f || !a.w2f || !a.w1_inv || !a.b1 || !a.w2_inv || !a.b2 || !a.y_bound || !a.q || !a.out || !a.gate || !a.ptr ||
      !a.tile_ptr || !a.ntiles || !a.tile_info || !a.batch)
    return ISG_EINVAL;                         // the struct the kernel dereferences, not the parameters it was filled from
  isg::readout_tile_kernel<<<(unsigned)max_tiles, 256, isg::RO_SMEM_BYTES, isg::as_stream(stream)>>>(a);
  return isg::check_launch();
}
