// Graph-aligned row tiles: the dense back half of an MGAT layer as ONE kernel.
//
// Reference: ISubGVQA/models/mgat.py:156-177 and the first line of the NEXT layer's convolution, mgat_v2_conv.py:156-157:
//     c = x_proj[i](conv_out)            Linear(H*C -> C*H/2) GELU Linear(C*H/2 -> C) GELU          mgat.py:156
//     c = scatter_scaled_dot_product_attention(ins_i, c, c, batch)                                   mgat.py:168
//     c = GraphNorm_i(c);  h = c + h;  [h = h * mask]                                                 mgat.py:171-177
//     x' = gelu(h * ins_{i+1}[batch])                                                                 mgat_v2_conv.py:156-157
// Un-fused that is five launches (two exact-split Linears, the per-graph tail, the instruction gate) and four HBM round
// trips of [N, .] tensors (the 256-wide intermediate, c, h, x').  The per-graph reductions of the tail need whole graphs,
// the GEMMs need 64-row panels: so the M-tile of this kernel is a RUN OF WHOLE GRAPHS with at most 64 nodes (~3 graphs at
// BASELINE configs[1]), found once per batch by isg_tile_plan.  Per tile:
//   GEMM1  [64 x 512] x W0^T -> [64 x 256]   A streamed in four 128-wide K chunks: global -> registers -> row scale ->
//          (hi, mid) fp16 planes in LDS (double buffered, one barrier per chunk); W0 never touches LDS: fragment-major
//          planes (isg_split_f16x2_frag) from L2 straight into the MFMA registers, two k-steps ahead; a wave owns all
//          64 rows x 64 columns, three v_mfma_f32_32x32x16_f16 per product (isg_gemm_f16x3.hip has the numerics)
//   GELU, row maxima across the four waves -> the intermediate as (hi, mid) planes in LDS (over the A buffers)
//   GEMM2  [64 x 256] x W2^T -> [64 x 128]   both K halves in their own accumulators
//   GELU -> c [64 x 128] fp32 in LDS; then the tail exactly as isg_norm_pool.hip::graph_tail_kernel<2> does it, on LDS
//   rows: wave per node for <ins_g, c_n> / sqrt(C), wave per graph for the softmax, thread per channel walking a graph's
//   nodes IN ORDER with unfused mul + add (the CPU scatter kernels' order and roundings), + h, * mask; the result and, when
//   a next layer exists, gelu(result * ins_next[g]) are written out.
// HBM traffic per layer at configs[1]: conv_out 168 MB + h 42 MB in, h' and x' 84 MB out (un-fused chain: ~590 MB).
#include "isg_f16x3.hpp"

#ifdef ISG_DT_STAMP
// Diagnostic build (tools/stamp_dense_tail.py): every wave records the core clock (s_memtime) at phase boundaries and writes
// the differences to a buffer of its own, [tile * 4 + wave][16]; no output value depends on a stamp.
static __device__ long long *g_dt_stamps = nullptr;
#define DT_T() ((long long)__builtin_amdgcn_s_memtime())
#define DT_STAMP(i) { const long long now_ = DT_T(); st_acc[i] = now_ - st_last; st_last = now_; }
#else
#define DT_STAMP(i)
#endif

namespace isg {

// =====================================================================================================================
// Tile plan: greedy packing of consecutive graphs into tiles of at most `ncap` nodes (and `ecap` CSR slots).
// One workgroup; the graphs are taken in chunks of 1024 (a chunk boundary closes a tile).  Inside a chunk
// next[g] = first graph of the tile after the one that STARTS at g (binary search over ptr / eptr); the tile starts are the
// orbit of the chunk's first graph under `next`, marked by pointer doubling in LDS (10 rounds), then compacted in order.
// =====================================================================================================================
constexpr int TP_CH = 1024;

__global__ __launch_bounds__(TP_CH) void tile_plan_kernel(const int *__restrict__ ptr, const int *__restrict__ eptr, int B,
                                                          int ncap, int ecap, int *__restrict__ tile_ptr,
                                                          int *__restrict__ ntiles, int cap) {
  __shared__ int s_jump[2][TP_CH];
  __shared__ int s_mark[TP_CH];
  __shared__ int s_wsum[TP_CH / 64];
  __shared__ int s_base;
  const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
  if (i == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < B; c0 += TP_CH) {
    const int cn = min(TP_CH, B - c0);
    int nx = TP_CH;
    if (i < cn) {
      const int g = c0 + i;
      const int nlim = ptr[g] + ncap;
      const int elim = eptr ? eptr[g] + ecap : 0;
      int lo = g + 1, hi = c0 + cn;      // the answer lies in [lo, hi]: the predicate is monotone, g + 1 is forced
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const bool ok = ptr[mid] <= nlim && (!eptr || eptr[mid] <= elim);
        if (ok) lo = mid; else hi = mid - 1;
      }
      nx = lo - c0 >= cn ? TP_CH : lo - c0;
    }
    s_jump[0][i] = nx;
    s_mark[i] = i == 0 ? 1 : 0;
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int step = 0; step < 10; ++step) {
      const int j = s_jump[cur][i];
      const int m = s_mark[i];
      const int jj = j < TP_CH ? s_jump[cur][j] : TP_CH;
      __syncthreads();
      if (m && j < TP_CH) s_mark[j] = 1;
      s_jump[cur ^ 1][i] = jj;
      __syncthreads();
      cur ^= 1;
    }
    const int m = i < cn ? s_mark[i] : 0;
    const unsigned long long bal = __ballot(m);
    if (lane == 0) s_wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base, total = 0;
    for (int w = 0; w < TP_CH / 64; ++w) {
      if (w < wave) off += s_wsum[w];
      total += s_wsum[w];
    }
    if (m) {
      const int idx = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (idx < cap) tile_ptr[idx] = c0 + i;
    }
    __syncthreads();
    if (i == 0) s_base += total;
    __syncthreads();
  }
  if (i == 0) {
    const int T = min(s_base, cap);
    tile_ptr[T] = B;
    *ntiles = T;
  }
}

// =====================================================================================================================
// The fused dense tail
// =====================================================================================================================
constexpr int DT_ROWS = 64, DT_KC = 128, DT_K1 = 512, DT_MID = 256, DT_C = 128;
constexpr int DT_LDA = DT_KC + 8, DT_LDY = DT_MID + 8, DT_LDC = DT_C + 4;
constexpr int DT_KS1 = DT_K1 / 16, DT_KS2 = DT_MID / 16;
constexpr int DT_BUF_BYTES = 2 * 2 * DT_ROWS * DT_LDA * 2;                 // 69,632: the two A buffers (and what aliases them)
constexpr int DT_GST = 8;                        // graphs of a tile whose instruction rows are staged in LDS (the rest: global)
constexpr int DT_SMEM_BYTES = DT_BUF_BYTES + (3 * 64 + 4 * 64 + 64 + 64 + 64 + 2 * DT_GST * DT_C) * 4;      // 80,384
static_assert(2 * DT_ROWS * DT_LDY * 2 <= DT_BUF_BYTES && DT_ROWS * DT_LDC * 4 + 2 * 32 * DT_C * 4 <= DT_BUF_BYTES, "aliases must fit");
static_assert(2 * DT_SMEM_BYTES <= 160 * 1024, "two workgroups per CU");

struct DtArgs {
  const float *a;          // conv output [N, 512], row stride lda
  const float *a_rowmax;   // [N, P] partial maxima of |a| per row, row stride ldp
  const _Float16 *w1f;     // x_proj.0 weight [256, 512] as fragment-major (hi, mid) planes
  const float *w1_inv, *b1;
  const float *y_bound;    // [2]: max_j sum_k |W0[j, k]| and max_j |b0[j]| -- |x_proj.0(a)_ij| <= amax_i * y_bound[0] + y_bound[1]
  const _Float16 *w2f;     // x_proj.2 weight [128, 256]
  const float *w2_inv, *b2;
  const float *ins, *h;    // [B, 128], [N, 128]
  const float *gn_w, *gn_b, *gn_ms;
  const float *node_mask, *ins_next;
  float *h_out, *xg_out;
  const int *ptr, *tile_ptr, *ntiles;
  const long long *batch;
  int N, lda, P, ldp;
  float eps, denom;
};

__global__ __launch_bounds__(256, 2) void mgat_dense_tail_kernel(DtArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dt_smem[];
  typedef _Float16 (*BufA)[2][DT_ROWS][DT_LDA];       // [buffer][plane][row][k]
  typedef _Float16 (*BufY)[DT_ROWS][DT_LDY];          // [plane][row][k]
  typedef float (*BufC)[DT_LDC];                      // [row][channel]
  BufA bufA = reinterpret_cast<BufA>(dt_smem);
  BufY sY = reinterpret_cast<BufY>(dt_smem);
  BufC sC = reinterpret_cast<BufC>(dt_smem);
  float *s_f = reinterpret_cast<float *>(dt_smem + DT_BUF_BYTES);
  float *s_inv1 = s_f, *s_scale2 = s_f + 64, *s_inv2 = s_f + 128, *s_rmax = s_f + 192, *s_a = s_f + 448;
  int *s_gid = reinterpret_cast<int *>(s_f + 512);
  float *s_mask = s_f + 576;
  float *s_ins = s_f + 640, *s_insn = s_f + 640 + DT_GST * DT_C;       // [DT_GST][C] each

  const int t = blockIdx.x;
  if (t >= *a.ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ISG_DT_STAMP
  long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long st_begin = DT_T();
  long long st_last = st_begin;
#endif
  const int fr = lane & 31, hh = lane >> 5, fk = hh * 8;
  const int g0 = a.tile_ptr[t], g1 = a.tile_ptr[t + 1];
  const int r0 = a.ptr[g0];
  const int nrows = min(a.ptr[g1] - r0, DT_ROWS);
  if (nrows <= 0) return;

  // ---- staging map: 8 float4 per thread and chunk; the 32 lanes of a half-wave hold one 512-byte row piece -----------------
  const int srow = tid >> 5, sc4 = tid & 31;          // rows srow + 8 u
#define DT_LOAD_CHUNK(c)                                                                                         \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                \
    const int row = srow + 8 * u;                                                                                \
    const int gr = min(r0 + min(row, nrows - 1), a.N - 1);                                                       \
    ra[u] = *reinterpret_cast<const float4 *>(a.a + (int64_t)gr * a.lda + (c) * DT_KC + sc4 * 4);                \
  }
  float4 ra[8];
  DT_LOAD_CHUNK(0)               // in flight under the tile's bookkeeping
  if (tid < DT_ROWS) {     // the row's scale from the producer's partial maxima: one round of loads per tile
    const int gr = min(r0 + min(tid, nrows - 1), a.N - 1);
    const float *rm = a.a_rowmax + (int64_t)gr * a.ldp;
    float mx = 0.f;
    if (a.P == 4) {
      mx = fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
    } else {
      for (int p = 0; p < a.P; ++p) mx = fmaxf(mx, rm[p]);
    }
    float s, inv;
    h3_scale(mx, s, inv);
    s_rmax[tid] = s;                    // a strip nothing else uses
    s_inv1[tid] = inv;
    // The intermediate's row scale needs no pass over it: |gelu(z)| <= |z| and |z_ij| <= amax_i * max_j ||W0_j||_1 + max |b0|.
    // The bound is loose by 2^6 or so, which costs nothing: an element's planes carry it to max(2^-22 |y|, 2^-25 / scale), and
    // even a row maximum scaled to 2^7 instead of 2^13 keeps that floor at 2^-32 of the row maximum (fp32 eps: 2^-24)
    h3_scale(fmaf(mx, a.y_bound[0], a.y_bound[1]), s, inv);
    s_scale2[tid] = s;
    s_inv2[tid] = inv;
    s_gid[tid] = (int)a.batch[gr];
    s_mask[tid] = a.node_mask ? a.node_mask[gr] : 1.f;
  }
  const int ng = g1 - g0;
  {   // the instruction rows of the tile's first DT_GST graphs (this layer's and the next one's): 32 lanes per row
    const int gi = tid >> 5, c4 = tid & 31;
    if (gi < min(ng, DT_GST)) {
      *reinterpret_cast<float4 *>(&s_ins[gi * DT_C + c4 * 4]) =
          *reinterpret_cast<const float4 *>(a.ins + (int64_t)(g0 + gi) * DT_C + c4 * 4);
      if (a.ins_next)
        *reinterpret_cast<float4 *>(&s_insn[gi * DT_C + c4 * 4]) =
            *reinterpret_cast<const float4 *>(a.ins_next + (int64_t)(g0 + gi) * DT_C + c4 * 4);
    }
  }
  __syncthreads();
  float sa[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) sa[u] = s_rmax[srow + 8 * u];
#define DT_WRITE_CHUNK(b)                                                                                        \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                \
    const int row = srow + 8 * u;                                                                                \
    float4 v = ra[u];                                                                                            \
    if (row >= nrows) v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
    v.x *= sa[u]; v.y *= sa[u]; v.z *= sa[u]; v.w *= sa[u];                                                      \
    hf16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};                                    \
    hf16x4 mid = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]), \
                  (_Float16)(v.w - (float)hi[3])};                                                               \
    *reinterpret_cast<hf16x4 *>(&bufA[b][0][row][sc4 * 4]) = hi;                                                 \
    *reinterpret_cast<hf16x4 *>(&bufA[b][1][row][sc4 * 4]) = mid;                                                \
  }
  DT_STAMP(0)                  // tile header + row scales
  DT_WRITE_CHUNK(0)
  DT_LOAD_CHUNK(1)

  // ---- GEMM1: [64 x 512] . W0^T, this wave's 64 columns (tiles 2 wave, 2 wave + 1) -------------------------------------------
  constexpr unsigned plane1 = (unsigned)(DT_MID / 32) * DT_KS1 * 1024u;
  const __amdgpu_buffer_rsrc_t wr1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.w1f), 0, (int)(2u * plane1), 0x00020000);
  const int voff = lane * 16;
  hf32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  hf16x8 wq[3][2][2], af[2][2][2];
#define DT_LOADW1(st, s)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
      wq[st][j][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(                           \
          wr1, voff, (int)(((unsigned)(2 * wave + j) * DT_KS1 + (unsigned)(s)) * 1024u + q * plane1), 0));
#define DT_LOADA1(st, b, ksl)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
      af[st][i][q] = *reinterpret_cast<const hf16x8 *>(&bufA[b][q][i * 32 + fr][(ksl) * 16 + fk]);
  // small terms first; the four accumulators take turns so that dependent MFMAs are four issues apart
#define DT_MMA1(A, W)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i][0], W[j][1], acc[i][j], 0, 0, 0);                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i][1], W[j][0], acc[i][j], 0, 0, 0);                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i][0], W[j][0], acc[i][j], 0, 0, 0);
  DT_LOADW1(0, 0)
  DT_LOADW1(1, 1)
  __syncthreads();                    // chunk 0 is in bufA[0]
  DT_STAMP(1)                  // chunk 0 staged
  DT_LOADA1(0, 0, 0)
#pragma unroll
  for (int s = 0; s < DT_KS1; ++s) {
    const int c = s >> 3, ksl = s & 7;
    if (s + 2 < DT_KS1) { DT_LOADW1((s + 2) % 3, s + 2) }
    if (ksl < 7) { DT_LOADA1((s + 1) & 1, c & 1, ksl + 1) }
    __builtin_amdgcn_sched_barrier(0);      // the prefetches stay AHEAD of this step's MFMAs (hipcc sinks them to their use otherwise)
    DT_MMA1(af[s & 1], wq[s % 3])
    __builtin_amdgcn_sched_barrier(0);
    if (ksl == 7 && c < 3) {
      DT_WRITE_CHUNK((c + 1) & 1)                  // last read two chunks ago: every wave is past that barrier
      if (c + 2 < 4) { DT_LOAD_CHUNK(c + 2) }
      __syncthreads();
      DT_STAMP(2 + c)          // chunk c computed, chunk c + 1 staged
      DT_LOADA1((s + 1) & 1, (c + 1) & 1, 0)
    }
  }
#ifdef ISG_DT_STAMP
  asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[1][1][15]));
#endif
  DT_STAMP(5)                  // last chunk computed
#undef DT_LOADW1
#undef DT_LOADA1
#undef DT_MMA1
#undef DT_LOAD_CHUNK
#undef DT_WRITE_CHUNK

  // ---- epilogue 1: scale back, + bias, GELU -> the intermediate's (hi, mid) planes (row scales: see the tile header) ----------
  {
    float wi[2], bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (2 * wave + j) * 32 + fr;
      wi[j] = a.w1_inv[col];
      bv[j] = a.b1[col];
    }
    __syncthreads();          // every wave is done with the A buffers: the planes may overwrite them
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float si = s_inv1[row], s2 = s_scale2[row];
        // both scales are powers of two: exact
        const isg_f32x2 v2 = gelu_exact2(isg_f32x2{(acc[i][0][r] * si) * wi[0] + bv[0], (acc[i][1][r] * si) * wi[1] + bv[1]}) * s2;
        const _Float16 h0 = (_Float16)v2.x, h1 = (_Float16)v2.y;
        sY[0][row][(2 * wave) * 32 + fr] = h0;
        sY[0][row][(2 * wave + 1) * 32 + fr] = h1;
        sY[1][row][(2 * wave) * 32 + fr] = (_Float16)(v2.x - (float)h0);
        sY[1][row][(2 * wave + 1) * 32 + fr] = (_Float16)(v2.y - (float)h1);
      }
  }
  __syncthreads();
  DT_STAMP(6)                  // epilogue 1: GELU, row maxima, planes of the intermediate

  // ---- GEMM2: [64 x 256] . W2^T, this wave's 32 columns; the two K halves in their own accumulators --------------------------
  constexpr unsigned plane2 = (unsigned)(DT_C / 32) * DT_KS2 * 1024u;
  const __amdgpu_buffer_rsrc_t wr2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.w2f), 0, (int)(2u * plane2), 0x00020000);
  hf32x4 rh[8];              // the tile's residual rows: in flight under GEMM2, used by the last pass of the tail
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int gr = min(r0 + min(srow + 8 * u, nrows - 1), a.N - 1);
    rh[u] = *reinterpret_cast<const hf32x4 *>(a.h + (int64_t)gr * DT_C + sc4 * 4);
  }
  hf32x16 acc2[2][2];        // [k half][row block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  {
    hf16x8 w2[4][2], a2[2][2][2];
#define DT_LOADW2(st, s)                                                                                         \
  _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                  \
      w2[st][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(                              \
          wr2, voff, (int)(((unsigned)wave * DT_KS2 + (unsigned)(s)) * 1024u + q * plane2), 0));
#define DT_LOADA2(st, s)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
      a2[st][i][q] = *reinterpret_cast<const hf16x8 *>(&sY[q][i * 32 + fr][(s) * 16 + fk]);
    DT_LOADW2(0, 0)
    DT_LOADW2(1, 1)
    DT_LOADW2(2, 2)
    DT_LOADA2(0, 0)
#pragma unroll
    for (int s = 0; s < DT_KS2; ++s) {
      const int kh = s & 1;             // alternate the two accumulator sets: dependent MFMAs are four issues apart
      if (s + 3 < DT_KS2) { DT_LOADW2((s + 3) & 3, s + 3) }
      if (s + 1 < DT_KS2) { DT_LOADA2((s + 1) & 1, s + 1) }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc2[kh][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[s & 1][i][0], w2[s & 3][1], acc2[kh][i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc2[kh][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[s & 1][i][1], w2[s & 3][0], acc2[kh][i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc2[kh][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[s & 1][i][0], w2[s & 3][0], acc2[kh][i], 0, 0, 0);
    }
#undef DT_LOADW2
#undef DT_LOADA2
  }
  const int col2 = wave * 32 + fr;
  const float wi2 = a.w2_inv[col2], bv2 = a.b2[col2];
#ifdef ISG_DT_STAMP
  asm volatile("" ::"v"(acc2[0][0][0]), "v"(acc2[1][1][15]));
#endif
  DT_STAMP(7)                  // GEMM2
  __syncthreads();          // every wave is done with the planes of the intermediate: c may overwrite them
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;        // r even: r + 1 is the next row
      const isg_f32x2 v2 = gelu_exact2(isg_f32x2{((acc2[0][i][r] + acc2[1][i][r]) * s_inv2[row]) * wi2 + bv2,
                                                 ((acc2[0][i][r + 1] + acc2[1][i][r + 1]) * s_inv2[row + 1]) * wi2 + bv2});
      sC[row][col2] = v2.x;
      sC[row + 1][col2] = v2.y;
    }
  __syncthreads();
  DT_STAMP(8)                  // epilogue 2

  // ---- the layer tail on the tile's graphs (isg_norm_pool.hip::graph_tail_kernel<2>, same arithmetic, rows from LDS) --------
  // phase A: a_n = <ins_g, c_n> / sqrt(C).  A half-wave per node, one float4 per lane, the 32-lane butterfly -- the same bits as
  // graph_tail_kernel's 64-lane wave_sum, whose upper half adds zeros; eight nodes per wave and iteration (independent chains)
  for (int kb = 8 * wave; kb < nrows; kb += 32) {
    float part[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(kb + 2 * u + hh, nrows - 1);
      const int gi = s_gid[k] - g0;
      const float4 q = gi < DT_GST ? *reinterpret_cast<const float4 *>(&s_ins[gi * DT_C + fr * 4])
                                   : *reinterpret_cast<const float4 *>(a.ins + (int64_t)(g0 + gi) * DT_C + fr * 4);
      const float4 v = *reinterpret_cast<const float4 *>(&sC[k][fr * 4]);
      part[u] = 0.f + dot4(v, q);
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
  DT_STAMP(9)                  // phase A
  // phase B: softmax over a graph's nodes (the sum runs in node order), a wave per graph
  for (int gi = wave; gi < ng; gi += 4) {
    const int nb = a.ptr[g0 + gi] - r0;
    const int n = min(a.ptr[g0 + gi + 1] - r0, nrows) - nb;
    if (n <= 0) continue;
    float *sa_g = s_a + nb;
    float mx = lane < n ? sa_g[lane] : -INFINITY;
    mx = wave_max(mx);
    if (lane < n) sa_g[lane] = expf(sa_g[lane] - mx);
    __builtin_amdgcn_wave_barrier();
    float sum = 0.f;
    for (int k = 0; k < n; ++k) sum += sa_g[k];
    sum += 0.f;
    __builtin_amdgcn_wave_barrier();
    if (lane < n) sa_g[lane] = sa_g[lane] / sum;
  }
  __syncthreads();
  DT_STAMP(10)                 // phase B
  // phase C: GraphNorm statistics by thread = (channel, every second graph), walking the graph's nodes IN ORDER with unfused
  // mul + add (the CPU scatter kernels' order and roundings) -> LDS; then the elementwise pass in the staging layout (thread =
  // 8 rows x 4 channels, the residual rows h still in its registers): + h, * mask, next gate, 16-byte row stores
  float(*s_mean)[DT_C] = reinterpret_cast<float(*)[DT_C]>(dt_smem + DT_ROWS * DT_LDC * 4);      // [32][C] behind c
  float(*s_std)[DT_C] = s_mean + 32;
  for (int gb = 0; gb < ng; gb += 32) {          // 32 graphs of statistics fit behind c (more only with 1-2-node graphs)
    const int ge = min(ng, gb + 32);
    {
      const int ch = tid & (DT_C - 1);
      const float ms = a.gn_ms[ch];
      for (int gi = gb + (tid >> 7); gi < ge; gi += 2) {
        const int nb = a.ptr[g0 + gi] - r0;
        const int n = min(a.ptr[g0 + gi + 1] - r0, nrows) - nb;
        if (n <= 0) continue;
        const float cnt = (float)n;
        float sum = 0.f;
#pragma unroll 4
        for (int k = 0; k < n; ++k) sum = __fadd_rn(sum, __fmul_rn(s_a[nb + k], sC[nb + k][ch]));
        const float mean_ms = __fmul_rn(sum / cnt, ms);
        float sq = 0.f;
#pragma unroll 4
        for (int k = 0; k < n; ++k) {
          const float o = __fsub_rn(__fmul_rn(s_a[nb + k], sC[nb + k][ch]), mean_ms);
          sq = __fadd_rn(sq, __fmul_rn(o, o));
        }
        s_mean[gi - gb][ch] = mean_ms;
        s_std[gi - gb][ch] = sqrtf(__fadd_rn(sq / cnt, a.eps));
      }
    }
    __syncthreads();
    {
      const float4 w4 = *reinterpret_cast<const float4 *>(a.gn_w + sc4 * 4), b4 = *reinterpret_cast<const float4 *>(a.gn_b + sc4 * 4);
      const float wv[4] = {w4.x, w4.y, w4.z, w4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = srow + 8 * u;
        const int gi = s_gid[min(row, nrows - 1)] - g0;
        if (row >= nrows || gi < gb || gi >= ge) continue;
        const float an = s_a[row], mk = s_mask[row];
        const float4 c4 = *reinterpret_cast<const float4 *>(&sC[row][sc4 * 4]);
        const float4 m4 = *reinterpret_cast<const float4 *>(&s_mean[gi - gb][sc4 * 4]);
        const float4 d4 = *reinterpret_cast<const float4 *>(&s_std[gi - gb][sc4 * 4]);
        float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.ins_next)
          x4 = gi < DT_GST ? *reinterpret_cast<const float4 *>(&s_insn[gi * DT_C + sc4 * 4])
                           : *reinterpret_cast<const float4 *>(a.ins_next + (int64_t)(g0 + gi) * DT_C + sc4 * 4);
        const float cv[4] = {c4.x, c4.y, c4.z, c4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
        hf32x4 y4, g4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float o = __fsub_rn(__fmul_rn(an, cv[j]), mv[j]);
          float y = __fadd_rn(__fmul_rn(wv[j], o) / dv[j], bv[j]);
          y = __fadd_rn(y, rh[u][j]);
          if (a.node_mask) y = __fmul_rn(mk, y);
          y4[j] = y;
        }
        if (a.xg_out) {
          const isg_f32x2 ga = gelu_exact2(isg_f32x2{y4[0] * xv[0], y4[1] * xv[1]});
          const isg_f32x2 gb2 = gelu_exact2(isg_f32x2{y4[2] * xv[2], y4[3] * xv[3]});
          g4 = hf32x4{ga.x, ga.y, gb2.x, gb2.y};
        }
        const int64_t at = (int64_t)(r0 + row) * DT_C + sc4 * 4;
        *reinterpret_cast<hf32x4 *>(a.h_out + at) = y4;
        if (a.xg_out) *reinterpret_cast<hf32x4 *>(a.xg_out + at) = g4;
      }
    }
    if (gb + 32 < ng) __syncthreads();
  }
#ifdef ISG_DT_STAMP
  DT_STAMP(11)                 // phase C
  if (g_dt_stamps && lane == 0) {
    st_acc[12] = DT_T() - st_begin;
    st_acc[13] = nrows;
    st_acc[14] = ng;
    long long *dst = g_dt_stamps + ((long long)t * 4 + wave) * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i] = st_acc[i];
  }
#endif
}

}  // namespace isg

using namespace isg;

#ifdef ISG_DT_STAMP
extern "C" int isg_dt_set_stamp_buffer(long long *buf) {      // diagnostic build only: [tiles * 4 waves][16] int64
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dt_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t isg_tile_plan_capacity(int64_t N, int64_t E, int64_t B, int32_t node_cap, int32_t edge_cap) {
  if (N < 0 || E < 0 || B < 0 || node_cap <= 0) return 0;
  // two consecutive tiles of a chunk together exceed one of the caps (greedy), every chunk may end on a short tile
  int64_t t = 2 * (N / node_cap + 1) + (B + TP_CH - 1) / TP_CH + 1;
  if (edge_cap > 0) t += 2 * (E / edge_cap + 1);
  return t < B ? t : B;
}

extern "C" int isg_tile_plan(const int32_t *ptr, const int32_t *eptr, int64_t B, int32_t node_cap, int32_t edge_cap,
                             int32_t *tile_ptr, int32_t *ntiles, int64_t capacity, void *stream) {
  if (B < 0 || node_cap <= 0 || capacity < 0 || !tile_ptr || !ntiles || (B > 0 && !ptr)) return ISG_EINVAL;
  if (eptr && edge_cap <= 0) return ISG_EINVAL;
  if (B >= (1ll << 31) || capacity >= (1ll << 31)) return ISG_EUNSUPPORTED;
  tile_plan_kernel<<<1, TP_CH, 0, as_stream(stream)>>>(ptr, eptr, (int)B, node_cap, edge_cap, tile_ptr, ntiles, (int)capacity);
  return check_launch();
}

extern "C" int isg_mgat_dense_tail(const float *conv_out, int32_t lda, const float *a_rowmax, int32_t P, int32_t ldp,
                                   const uint16_t *w1_frag, const float *w1_inv_scale, const float *b1, const float *y_bound,
                                   const uint16_t *w2_frag, const float *w2_inv_scale, const float *b2, const float *ins,
                                   const float *h, const float *gn_weight, const float *gn_bias, const float *gn_mean_scale,
                                   double eps, const float *node_mask, const float *ins_next, float *h_out, float *xg_out,
                                   const int32_t *ptr, const int64_t *batch, const int32_t *tile_ptr, const int32_t *ntiles,
                                   int64_t max_tiles, int64_t N, int32_t K1, int32_t MID, int32_t C, void *stream) {
  if (N < 0 || max_tiles < 0 || lda < K1 || P <= 0 || ldp < P) return ISG_EINVAL;
  if (K1 != DT_K1 || MID != DT_MID || C != DT_C || (lda & 3) != 0 || N >= (1ll << 31) || max_tiles >= (1ll << 31) ||
      (reinterpret_cast<uintptr_t>(conv_out) & 15) != 0 || (reinterpret_cast<uintptr_t>(ins) & 15) != 0)
    return ISG_EUNSUPPORTED;
  if (N == 0 || max_tiles == 0) return ISG_OK;
  if (!conv_out || !a_rowmax || !w1_frag || !w1_inv_scale || !b1 || !y_bound || !w2_frag || !w2_inv_scale || !b2 || !ins || !h ||
      !gn_weight || !gn_bias || !gn_mean_scale || !h_out || !ptr || !batch || !tile_ptr || !ntiles)
    return ISG_EINVAL;
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&mgat_dense_tail_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, DT_SMEM_BYTES) == hipSuccess;
  if (!ok) return ISG_EUNSUPPORTED;
  DtArgs a;
  a.a = conv_out; a.a_rowmax = a_rowmax; a.w1f = reinterpret_cast<const _Float16 *>(w1_frag); a.w1_inv = w1_inv_scale;
  a.b1 = b1; a.y_bound = y_bound; a.w2f = reinterpret_cast<const _Float16 *>(w2_frag); a.w2_inv = w2_inv_scale; a.b2 = b2; a.ins = ins; a.h = h;
  a.gn_w = gn_weight; a.gn_b = gn_bias; a.gn_ms = gn_mean_scale; a.node_mask = node_mask; a.ins_next = ins_next;
  a.h_out = h_out; a.xg_out = xg_out; a.ptr = ptr; a.tile_ptr = tile_ptr; a.ntiles = ntiles;
  a.batch = reinterpret_cast<const long long *>(batch);
  a.N = (int)N; a.lda = lda; a.P = P; a.ldp = ldp; a.eps = (float)eps; a.denom = (float)sqrt((double)DT_C);
  mgat_dense_tail_kernel<<<(unsigned)max_tiles, 256, DT_SMEM_BYTES, as_stream(stream)>>>(a);
  return check_launch();
}
