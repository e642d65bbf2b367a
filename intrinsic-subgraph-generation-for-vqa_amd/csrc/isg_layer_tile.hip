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

#include "isg_diag.hpp"

ISG_DIAG_BUFFER(g_dt_stamps)            // -DISG_DIAG builds only (tools/stamp_dense_tail.py, stamp_tile_conv.py): [tile * 4 + wave][16] int64
#define DT_STAMP(i) ISG_DIAG_SET(i)     // the dense tail: one pass per launch
#define TC_STAMP(i) ISG_DIAG_ADD(i)     // the tile convolution: persistent, accumulated over its tiles

namespace isg {

// =====================================================================================================================
// Tile plan: greedy packing of consecutive graphs into tiles of at most `ncap` nodes (and `ecap` CSR slots).
// One workgroup; the graphs are taken in chunks of 1024 (a chunk boundary closes a tile).  Inside a chunk
// next[g] = first graph of the tile after the one that STARTS at g (binary search over ptr / eptr); the tile starts are the
// orbit of the chunk's first graph under `next`, marked by pointer doubling in LDS (10 rounds), then compacted in order.
// =====================================================================================================================
constexpr int TP_CH = 1024;

__device__ __forceinline__ void tile_plan_body(const int *__restrict__ ptr, const int *__restrict__ eptr, int B, int ncap, int ecap,
                                               int *__restrict__ tile_ptr, int *__restrict__ ntiles, int cap,
                                               int4 *__restrict__ tile_info, int4 *__restrict__ tile_heavy_first = nullptr) {
  __shared__ int s_jump[2][TP_CH];
  __shared__ int s_mark[TP_CH];
  __shared__ int s_ptr[TP_CH + 1], s_eptr[TP_CH + 1];
  __shared__ int s_wsum[TP_CH / 64];
  __shared__ int s_base;
  const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
  if (i == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < B; c0 += TP_CH) {
    const int cn = min(TP_CH, B - c0);
    // the chunk's node / slot offsets in LDS: the binary search below is 11 dependent reads per graph (from global memory that
    // was ~1 us each: most of this kernel's 21 us)
    for (int k = i; k <= cn; k += TP_CH) {
      s_ptr[k] = ptr[c0 + k];
      s_eptr[k] = eptr ? eptr[c0 + k] : 0;
    }
    __syncthreads();
    int nx = TP_CH;
    if (i < cn) {
      const int nlim = s_ptr[i] + ncap;
      const int elim = s_eptr[i] + ecap;
      int lo = i + 1, hi = cn;           // the answer lies in [lo, hi]: the predicate is monotone, i + 1 is forced
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const bool ok = s_ptr[mid] <= nlim && (!eptr || s_eptr[mid] <= elim);
        if (ok) lo = mid; else hi = mid - 1;
      }
      nx = lo >= cn ? TP_CH : lo;
    }
    const int nx0 = nx;                  // the tile that starts at this graph ends before graph c0 + nx0 (or with the chunk)
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
      if (idx < cap) {
        tile_ptr[idx] = c0 + i;
        if (tile_info) {       // {first node, nodes, first CSR slot, CSR slots}: one load gives a workgroup its tile
          const int l1 = nx0 < TP_CH ? nx0 : cn;
          int nn = s_ptr[l1] - s_ptr[i], ee = s_eptr[l1] - s_eptr[i];
          // A graph beyond the caps is a tile of its own (i + 1 is forced above).  It gets NO rows and NO slots here: the tile
          // kernels pass over it, and the caller runs the per-graph kernels on the list of such graphs (mixed dispatch:
          // ops.GraphPlan.oversize) -- one 130-node graph no longer switches a whole batch off the tile kernels.
          if (nn > ncap || (eptr && ee > ecap)) { nn = 0; ee = 0; }
          tile_info[idx] = make_int4(s_ptr[i], nn, s_eptr[i], ee);
        }
      }
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
  // The same descriptors HEAVY TILES FIRST (by 32-slot half chunks, the unit of the layer kernel's edge loop; ties in tile order):
  // a persistent kernel whose workgroup w walks entries w, w + G, w + 2G, ... of THIS list gets one tile of every weight class
  // per round instead of whatever the batch order deals it -- at BASELINE configs[1] the slowest workgroup's share of the work
  // drops from 1.062x to 1.029x the mean (tools/sim_tile_balance.py).  A stable counting sort over 9 classes on every wave of the
  // workgroup: one pass of ballots counts, one pass places 1024 descriptors per round (one wave doing both took 15 us of the
  // launch -- 48 dependent rounds of a global load and 9 ballots -- behind which the whole chip waited).
  if (tile_info && tile_heavy_first) {          // (uniform; every tile_info entry is written and behind a barrier)
    constexpr int NB = 9;                       // classes: 8, 7, ..., 0 half chunks (8 = a full 256-slot tile)
    constexpr int NW = TP_CH / 64;
    __shared__ int s_cnt[NW][NB];
    __shared__ int s_start[NB];
    __syncthreads();                            // s_base is final
    const int T = min(s_base, cap);
    int mine[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) mine[b] = 0;
    for (int t0 = 0; t0 < T; t0 += TP_CH) {      // every thread walks every round: the ballots need whole waves
      const int t = t0 + i;
      const int cls = t < T ? NB - 1 - min((tile_info[t].w + 31) >> 5, NB - 1) : -1;
#pragma unroll
      for (int b = 0; b < NB; ++b) mine[b] += __popcll(__ballot(cls == b));
    }
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) s_cnt[wave][b] = mine[b];
    }
    __syncthreads();
    if (i == 0) {
      int run = 0;
      for (int b = 0; b < NB; ++b) {
        int c = 0;
        for (int w = 0; w < NW; ++w) c += s_cnt[w][b];
        s_start[b] = run;
        run += c;
      }
    }
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += TP_CH) {      // 1024 descriptors per round, wave w the 64 at t0 + 64 w: tile order within a class
      const int t = t0 + i;
      int4 d = make_int4(0, 0, 0, 0);
      if (t < T) d = tile_info[t];
      const int cls = t < T ? NB - 1 - min((d.w + 31) >> 5, NB - 1) : -1;
      int below = 0, wcnt[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const unsigned long long m = __ballot(cls == b);
        wcnt[b] = __popcll(m);
        if (cls == b) below = __popcll(m & ((1ull << lane) - 1ull));
      }
      if (lane == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) s_cnt[wave][b] = wcnt[b];
      }
      __syncthreads();
      if (cls >= 0) {
        int at = s_start[cls] + below;
        for (int w = 0; w < wave; ++w) at += s_cnt[w][cls];
        tile_heavy_first[at] = d;
      }
      __syncthreads();
      if (i < NB) {
        int c = 0;
        for (int w = 0; w < NW; ++w) c += s_cnt[w][i];
        s_start[i] += c;
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(TP_CH) void tile_plan_kernel(const int *__restrict__ ptr, const int *__restrict__ eptr, int B,
                                                          int ncap, int ecap, int *__restrict__ tile_ptr,
                                                          int *__restrict__ ntiles, int cap, int4 *__restrict__ tile_info,
                                                          int4 *__restrict__ tile_heavy_first) {
  tile_plan_body(ptr, eptr, B, ncap, ecap, tile_ptr, ntiles, cap, tile_info, tile_heavy_first);
}

// =====================================================================================================================
// The fused dense tail
// =====================================================================================================================
constexpr int DT_ROWS = 64, DT_KC = 128, DT_K1 = 512, DT_MID = 256, DT_C = 128;
constexpr int DT_LDA = DT_KC + 8, DT_LDY = DT_MID + 8, DT_LDC = DT_C + 4;
constexpr int DT_KS1 = DT_K1 / 16, DT_KS2 = DT_MID / 16;
constexpr int DT_BUF_BYTES = 2 * 2 * DT_ROWS * DT_LDA * 2;                 // 69,632: the two A buffers (and what aliases them)
constexpr int DT_GPC = 128;                      // graphs of a tile whose node offsets are staged in LDS (the rest: global)
constexpr int DT_GST = 8;                        // graphs of a tile whose instruction rows are staged in LDS (the rest: global)
constexpr int DT_SMEM_BYTES = DT_BUF_BYTES + (3 * 64 + 4 * 64 + 64 + 64 + 64 + 2 * DT_GST * DT_C + DT_GPC + 4) * 4;      // 80,912
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
  const float *node_mask, *ins_next;    // optional (ins_next: required with xg_out / xp_out, checked by the wrapper)
  float *h_out;
  float *xg_out, *xinv_out;         // optional
  _Float16 *xp_out;                 // the gated rows as scaled (hi, mid) planes [N][2][128] for isg_gatv2_layer_conv (optional)
  const int *ptr, *tile_ptr, *ntiles;
  const int4 *tile_info;   // {first node, nodes, ., .} per tile: the rows' loads start one round trip after the launch
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
  const int4 tinfo = a.tile_info[t];        // requested together with the count (entries beyond it are never used)
  if (t >= *a.ntiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ISG_DIAG_BEGIN()
  const int fr = lane & 31, hh = lane >> 5, fk = hh * 8;
  const int r0 = tinfo.x;
  const int nrows = min(tinfo.y, DT_ROWS);
  if (nrows <= 0) return;
  const int g0 = a.tile_ptr[t], g1 = a.tile_ptr[t + 1];      // needed by the tail only: off the rows' critical path

  // ---- staging map: 8 float4 per thread and chunk; the 32 lanes of a half-wave hold one 512-byte row piece -----------------
  const int srow = tid >> 5, sc4 = tid & 31;          // rows srow + 8 u
#define DT_LOAD_CHUNK(c)                                                                                         \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                \
    const int row = srow + 8 * u;                                                                                \
    const int gr = min(r0 + min(row, nrows - 1), a.N - 1);                                                       \
    { const hf32x4 t_ = __builtin_nontemporal_load(reinterpret_cast<const hf32x4 *>(a.a + (int64_t)gr * a.lda + (c) * DT_KC + sc4 * 4)); ra[u] = make_float4(t_[0], t_[1], t_[2], t_[3]); } \
  }
  float4 ra[8];
  DT_LOAD_CHUNK(0)               // in flight under the tile's bookkeeping
  // Every other input of the header is requested HERE, unconditionally (clamped indices), before anything is used: a load inside
  // `if (tid < 64) { ... s_x[tid] = p[i]; }` is waited for inside its branch, and the header was eight such round trips in a row
  // (9 900 cycles of a workgroup's 90 000: profiles/r03_g_dense_tail_stamps.txt); now it is the one behind tile_info / tile_ptr.
  const int ng = g1 - g0;
  const int hrow = min(r0 + min(tid, nrows - 1), a.N - 1);          // the row whose scales thread tid < DT_ROWS files
  const float *rm = a.a_rowmax + (int64_t)hrow * a.ldp;
  float rmv[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) rmv[p] = rm[min(p, a.P - 1)];
  const int h_gid = (int)a.batch[hrow];
  float h_mask = 1.f;
  if (a.node_mask) h_mask = a.node_mask[hrow];                      // (wave-uniform: the pointer is a kernel argument)
  const int h_gp = a.ptr[g0 + min(tid, ng)];
  const int gi = tid >> 5, c4 = tid & 31;
  const int gi_c = min(gi, max(min(ng, DT_GST) - 1, 0));
  const float4 h_ins = *reinterpret_cast<const float4 *>(a.ins + (int64_t)(g0 + gi_c) * DT_C + c4 * 4);
  float4 h_insn = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.ins_next) h_insn = *reinterpret_cast<const float4 *>(a.ins_next + (int64_t)(g0 + gi_c) * DT_C + c4 * 4);
  if (tid < DT_ROWS) {     // the row's scale from the producer's partial maxima: one round of loads per tile
    float mx = fmaxf(fmaxf(rmv[0], rmv[1]), fmaxf(rmv[2], rmv[3]));      // (P < 4: the clamped reads repeat the last one)
    for (int p = 4; p < a.P; ++p) mx = fmaxf(mx, rm[p]);
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
    s_gid[tid] = h_gid;
    s_mask[tid] = h_mask;
  }
  int *s_gp = reinterpret_cast<int *>(s_f + 2688);      // node offsets of the tile's first DT_GPC graphs: the tail's loops read these
  if (tid <= min(ng, DT_GPC)) s_gp[tid] = h_gp - r0;
  // the instruction rows of the tile's first DT_GST graphs (this layer's and the next one's): 32 lanes per row
  if (gi < min(ng, DT_GST)) {
    *reinterpret_cast<float4 *>(&s_ins[gi * DT_C + c4 * 4]) = h_ins;
    if (a.ins_next) *reinterpret_cast<float4 *>(&s_insn[gi * DT_C + c4 * 4]) = h_insn;
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
  hf16x8 wq[4][2][2], af[2][2];      // W fragments three k-steps ahead; ONE set of panel fragments: a step's LDS reads are
                                     // issued behind the previous step's twelve MFMAs (384 cycles of matrix-core work)
#define DT_LOADW1(st, s)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
      wq[st][j][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(                           \
          wr1, voff, (int)(((unsigned)(2 * wave + j) * DT_KS1 + (unsigned)(s)) * 1024u + q * plane1), 0));
#define DT_LOADA1(b, ksl)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
      af[i][q] = *reinterpret_cast<const hf16x8 *>(&bufA[b][q][i * 32 + fr][(ksl) * 16 + fk]);
  // small terms first; the four accumulators take turns so that dependent MFMAs are four issues apart.  TRANSPOSED (W fragment =
  // the MFMA's A operand, the panel fragment its B operand; the fragments are the same either way): a lane then holds ONE row and
  // 16 columns of it in four runs of four, so the epilogue writes 8-byte plane pieces and reads a row's scales once -- with the
  // panel as the A operand a lane holds one column of 16 rows: 128 two-byte LDS stores and 64 scale reads per lane
#define DT_MMA1(A, W)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[j][1], A[i][0], acc[i][j], 0, 0, 0);                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[j][0], A[i][1], acc[i][j], 0, 0, 0);                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[j][0], A[i][0], acc[i][j], 0, 0, 0);
  DT_LOADW1(0, 0)
  DT_LOADW1(1, 1)
  DT_LOADW1(2, 2)
  __syncthreads();                    // chunk 0 is in bufA[0]
  DT_STAMP(1)                  // chunk 0 staged
#pragma unroll
  for (int s = 0; s < DT_KS1; ++s) {
    const int c = s >> 3, ksl = s & 7;
    if (s + 3 < DT_KS1) { DT_LOADW1((s + 3) & 3, s + 3) }
    DT_LOADA1(c & 1, ksl)
    __builtin_amdgcn_sched_barrier(0);      // the prefetches stay AHEAD of this step's MFMAs (hipcc sinks them to their use otherwise)
    DT_MMA1(af, wq[s & 3])
    __builtin_amdgcn_sched_barrier(0);
    if (ksl == 7 && c < 3) {
      DT_WRITE_CHUNK((c + 1) & 1)                  // last read two chunks ago: every wave is past that barrier
      if (c + 2 < 4) { DT_LOAD_CHUNK(c + 2) }
      __syncthreads();
      DT_STAMP(2 + c)          // chunk c computed, chunk c + 1 staged
    }
  }
  ISG_DIAG_KEEP2(acc[0][0][0], acc[1][1][15])
  DT_STAMP(5)                  // last chunk computed
#undef DT_LOADW1
#undef DT_LOADA1
#undef DT_MMA1
#undef DT_LOAD_CHUNK
#undef DT_WRITE_CHUNK

  // ---- epilogue 1: scale back, + bias, GELU -> the intermediate's (hi, mid) planes (row scales: see the tile header) ----------
  {
    // this lane's 2 x 4 runs of four columns: their W scales and biases (L2; the registers of the W fragments are free now)
    hf32x4 wi4[2][4], bv4[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = (2 * wave + j) * 32 + 8 * g + 4 * hh;
        wi4[j][g] = *reinterpret_cast<const hf32x4 *>(a.w1_inv + col);
        bv4[j][g] = *reinterpret_cast<const hf32x4 *>(a.b1 + col);
      }
    __syncthreads();          // every wave is done with the A buffers: the planes may overwrite them
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 32 + fr;
      const float si = s_inv1[row], s2 = s_scale2[row];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // both scales are powers of two: exact
          const isg_f32x2 va = gelu_exact2(isg_f32x2{(acc[i][j][4 * g] * si) * wi4[j][g][0] + bv4[j][g][0],
                                                     (acc[i][j][4 * g + 1] * si) * wi4[j][g][1] + bv4[j][g][1]}) * s2;
          const isg_f32x2 vb = gelu_exact2(isg_f32x2{(acc[i][j][4 * g + 2] * si) * wi4[j][g][2] + bv4[j][g][2],
                                                     (acc[i][j][4 * g + 3] * si) * wi4[j][g][3] + bv4[j][g][3]}) * s2;
          const hf16x4 hi = {(_Float16)va.x, (_Float16)va.y, (_Float16)vb.x, (_Float16)vb.y};
          const hf16x4 mid = {(_Float16)(va.x - (float)hi[0]), (_Float16)(va.y - (float)hi[1]), (_Float16)(vb.x - (float)hi[2]),
                              (_Float16)(vb.y - (float)hi[3])};
          const int col = (2 * wave + j) * 32 + 8 * g + 4 * hh;
          *reinterpret_cast<hf16x4 *>(&sY[0][row][col]) = hi;
          *reinterpret_cast<hf16x4 *>(&sY[1][row][col]) = mid;
        }
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
    rh[u] = __builtin_nontemporal_load(reinterpret_cast<const hf32x4 *>(a.h + (int64_t)gr * DT_C + sc4 * 4));      // read once
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
        acc2[kh][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s & 3][1], a2[s & 1][i][0], acc2[kh][i], 0, 0, 0);       // transposed,
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc2[kh][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s & 3][0], a2[s & 1][i][1], acc2[kh][i], 0, 0, 0);       // as GEMM1
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc2[kh][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s & 3][0], a2[s & 1][i][0], acc2[kh][i], 0, 0, 0);
    }
#undef DT_LOADW2
#undef DT_LOADA2
  }
  hf32x4 wi2[4], bv2[4];     // this lane's four runs of four columns
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    wi2[g] = *reinterpret_cast<const hf32x4 *>(a.w2_inv + wave * 32 + 8 * g + 4 * hh);
    bv2[g] = *reinterpret_cast<const hf32x4 *>(a.b2 + wave * 32 + 8 * g + 4 * hh);
  }
  ISG_DIAG_KEEP2(acc2[0][0][0], acc2[1][1][15])
  DT_STAMP(7)                  // GEMM2
  __syncthreads();          // every wave is done with the planes of the intermediate: c may overwrite them
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + fr;
    const float si2 = s_inv2[row];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const isg_f32x2 va = gelu_exact2(isg_f32x2{((acc2[0][i][4 * g] + acc2[1][i][4 * g]) * si2) * wi2[g][0] + bv2[g][0],
                                                 ((acc2[0][i][4 * g + 1] + acc2[1][i][4 * g + 1]) * si2) * wi2[g][1] + bv2[g][1]});
      const isg_f32x2 vb = gelu_exact2(isg_f32x2{((acc2[0][i][4 * g + 2] + acc2[1][i][4 * g + 2]) * si2) * wi2[g][2] + bv2[g][2],
                                                 ((acc2[0][i][4 * g + 3] + acc2[1][i][4 * g + 3]) * si2) * wi2[g][3] + bv2[g][3]});
      *reinterpret_cast<hf32x4 *>(&sC[row][wave * 32 + 8 * g + 4 * hh]) = hf32x4{va.x, va.y, vb.x, vb.y};
    }
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
      part[u] = 0.f + dot4_rn(v, q);
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
    const int nb = gi < DT_GPC ? s_gp[gi] : a.ptr[g0 + gi] - r0;
    const int n = min(gi < DT_GPC ? s_gp[gi + 1] : a.ptr[g0 + gi + 1] - r0, nrows) - nb;
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
        const int nb = gi < DT_GPC ? s_gp[gi] : a.ptr[g0 + gi] - r0;
        const int n = min(gi < DT_GPC ? s_gp[gi + 1] : a.ptr[g0 + gi + 1] - r0, nrows) - nb;
        if (n <= 0) continue;
        const float cnt = (float)n;
        float sum = 0.f;
#pragma unroll 4
        for (int k = 0; k < n; ++k) sum = add_rn(sum, mul_rn(s_a[nb + k], sC[nb + k][ch]));
        const float mean_ms = mul_rn(sum / cnt, ms);
        float sq = 0.f;
#pragma unroll 4
        for (int k = 0; k < n; ++k) {
          const float o = sub_rn(mul_rn(s_a[nb + k], sC[nb + k][ch]), mean_ms);
          sq = add_rn(sq, mul_rn(o, o));
        }
        s_mean[gi - gb][ch] = mean_ms;
        s_std[gi - gb][ch] = sqrtf(add_rn(sq / cnt, a.eps));
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
          const float o = sub_rn(mul_rn(an, cv[j]), mv[j]);
          float y = add_rn(mul_rn(wv[j], o) / dv[j], bv[j]);
          y = add_rn(y, rh[u][j]);
          if (a.node_mask) y = mul_rn(mk, y);
          y4[j] = y;
        }
        if (a.xg_out || a.xp_out) {
          const isg_f32x2 ga = gelu_exact2(isg_f32x2{y4[0] * xv[0], y4[1] * xv[1]});
          const isg_f32x2 gb2 = gelu_exact2(isg_f32x2{y4[2] * xv[2], y4[3] * xv[3]});
          g4 = hf32x4{ga.x, ga.y, gb2.x, gb2.y};
        }
        const int64_t at = (int64_t)(r0 + row) * DT_C + sc4 * 4;
        *reinterpret_cast<hf32x4 *>(a.h_out + at) = y4;
        if (a.xg_out) *reinterpret_cast<hf32x4 *>(a.xg_out + at) = g4;
        if (a.xp_out) {          // row scale + (hi, mid) split once per row here, not once per (tile, head) in the layer kernel
          const float mx = group_max<32>(fmaxf(fmaxf(fabsf(g4[0]), fabsf(g4[1])), fmaxf(fabsf(g4[2]), fabsf(g4[3]))));
          float sc, inv;
          h3_scale(mx, sc, inv);
          if (sc4 == 0) a.xinv_out[r0 + row] = inv;
          g4 *= sc;
          hf16x4 hi = {(_Float16)g4[0], (_Float16)g4[1], (_Float16)g4[2], (_Float16)g4[3]};
          hf16x4 mid = {(_Float16)(g4[0] - (float)hi[0]), (_Float16)(g4[1] - (float)hi[1]), (_Float16)(g4[2] - (float)hi[2]),
                        (_Float16)(g4[3] - (float)hi[3])};
          *reinterpret_cast<hf16x4 *>(a.xp_out + (int64_t)(r0 + row) * 256 + sc4 * 4) = hi;
          *reinterpret_cast<hf16x4 *>(a.xp_out + (int64_t)(r0 + row) * 256 + 128 + sc4 * 4) = mid;
        }
      }
    }
    if (gb + 32 < ng) __syncthreads();
  }
  DT_STAMP(11)                 // phase C
  ISG_DIAG_DUMP(g_dt_stamps, t * 4 + wave, 12, st_acc[13] = nrows; st_acc[14] = ng;)
}

}  // namespace isg

using namespace isg;

ISG_DIAG_SETTER(isg_dt_set_stamp_buffer, g_dt_stamps)

extern "C" int64_t isg_tile_plan_capacity(int64_t N, int64_t E, int64_t B, int32_t node_cap, int32_t edge_cap) {
  if (N < 0 || E < 0 || B < 0 || node_cap <= 0) return 0;
  // two consecutive tiles of a chunk together exceed one of the caps (greedy), every chunk may end on a short tile
  int64_t t = 2 * (N / node_cap + 1) + (B + TP_CH - 1) / TP_CH + 1;
  if (edge_cap > 0) t += 2 * (E / edge_cap + 1);
  return t < B ? t : B;
}

extern "C" int isg_tile_plan(const int32_t *ptr, const int32_t *eptr, int64_t B, int32_t node_cap, int32_t edge_cap,
                             int32_t *tile_ptr, int32_t *ntiles, int32_t *tile_info, int64_t capacity,
                             int32_t *tile_info_heavy_first, void *stream) {
  if (B < 0 || node_cap <= 0 || capacity < 0 || !tile_ptr || !ntiles || (B > 0 && !ptr) || (tile_info_heavy_first && !tile_info))
    return ISG_EINVAL;
  if (tile_info_heavy_first && (reinterpret_cast<uintptr_t>(tile_info_heavy_first) & 15) != 0) return ISG_EINVAL;
  if (eptr && edge_cap <= 0) return ISG_EINVAL;
  if (B >= (1ll << 31) || capacity >= (1ll << 31)) return ISG_EUNSUPPORTED;
  if (tile_info && (reinterpret_cast<uintptr_t>(tile_info) & 15) != 0) return ISG_EINVAL;
  tile_plan_kernel<<<1, TP_CH, 0, as_stream(stream)>>>(ptr, eptr, (int)B, node_cap, edge_cap, tile_ptr, ntiles, (int)capacity,
                                                       reinterpret_cast<int4 *>(tile_info),
                                                       reinterpret_cast<int4 *>(tile_info_heavy_first));
  return check_launch();
}

extern "C" int isg_mgat_dense_tail(const float *conv_out, int32_t lda, const float *a_rowmax, int32_t P, int32_t ldp,
                                   const uint16_t *w1_frag, const float *w1_inv_scale, const float *b1, const float *y_bound,
                                   const uint16_t *w2_frag, const float *w2_inv_scale, const float *b2, const float *ins,
                                   const float *h, const float *gn_weight, const float *gn_bias, const float *gn_mean_scale,
                                   double eps, const float *node_mask, const float *ins_next, float *h_out, float *xg_out,
                                   uint16_t *xp_out, float *xinv_out, const int32_t *ptr, const int64_t *batch,
                                   const int32_t *tile_ptr, const int32_t *tile_info, const int32_t *ntiles, int64_t max_tiles,
                                   int64_t N, int32_t K1, int32_t MID, int32_t C, void *stream) {
  if (N < 0 || max_tiles < 0 || lda < K1 || P <= 0 || ldp < P) return ISG_EINVAL;
  if (K1 != DT_K1 || MID != DT_MID || C != DT_C || (lda & 3) != 0 || N >= (1ll << 31) || max_tiles >= (1ll << 31) ||
      (reinterpret_cast<uintptr_t>(conv_out) & 15) != 0 || (reinterpret_cast<uintptr_t>(ins) & 15) != 0 ||
      ((reinterpret_cast<uintptr_t>(w1_inv_scale) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(w2_inv_scale) |
        reinterpret_cast<uintptr_t>(b2)) & 15) != 0)          // the epilogues read them as 16-byte runs
    return ISG_EUNSUPPORTED;
  if (N == 0 || max_tiles == 0) return ISG_OK;
  if (!conv_out || !a_rowmax || !w1_frag || !w1_inv_scale || !b1 || !y_bound || !w2_frag || !w2_inv_scale || !b2 || !ins || !h ||
      !gn_weight || !gn_bias || !gn_mean_scale || !h_out || !ptr || !batch || !tile_ptr || !tile_info || !ntiles ||
      ((xg_out || xp_out) && !ins_next) || (xp_out && !xinv_out))
    return ISG_EINVAL;
  if (!dyn_lds_ok<&mgat_dense_tail_kernel>(DT_SMEM_BYTES)) return ISG_EUNSUPPORTED;
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  DtArgs a = {
      .a = conv_out, .a_rowmax = a_rowmax, .w1f = reinterpret_cast<const _Float16 *>(w1_frag), .w1_inv = w1_inv_scale, .b1 = b1,
      .y_bound = y_bound, .w2f = reinterpret_cast<const _Float16 *>(w2_frag), .w2_inv = w2_inv_scale, .b2 = b2, .ins = ins, .h = h,
      .gn_w = gn_weight, .gn_b = gn_bias, .gn_ms = gn_mean_scale, .node_mask = node_mask, .ins_next = ins_next, .h_out = h_out,
      .xg_out = xg_out, .xinv_out = xinv_out, .xp_out = reinterpret_cast<_Float16 *>(xp_out), .ptr = ptr, .tile_ptr = tile_ptr,
      .ntiles = ntiles, .tile_info = reinterpret_cast<const int4 *>(tile_info), .batch = reinterpret_cast<const long long *>(batch),
      .N = (int)N, .lda = lda, .P = P, .ldp = ldp, .eps = (float)eps, .denom = (float)sqrt((double)DT_C)};
  if (!a.a || !a.a_rowmax || !a.w1f || !a.w1_inv || !a.b1 || !a.y_bound || !a.w2f || !a.w2_inv || !a.b2 || !a.ins || !a.h || !a.gn_w ||
      !a.gn_b || !a.gn_ms || !a.h_out || !a.ptr || !a.tile_ptr || !a.ntiles || !a.tile_info || !a.batch ||
      ((a.xg_out || a.xp_out) && !a.ins_next) || (a.xp_out && !a.xinv_out))
    return ISG_EINVAL;                         // the struct the kernel dereferences, not the parameters it was filled from
  mgat_dense_tail_kernel<<<(unsigned)max_tiles, 256, DT_SMEM_BYTES, as_stream(stream)>>>(a);
  return check_launch();
}

// =====================================================================================================================
// The convolution's message passing on the same tiles: edge GEMM -> logits -> softmax -> aggregation, one launch.
//
// Reference: MaskingGATv2Conv.message + aggregate, ISubGVQA/models/mgat_v2_conv.py:243-279 (with lin_edge, :259-261, inside).
// The pair it replaces (isg_gatv2_edge_logits + isg_gatv2_mp_fwd_logits, csrc/isg_mp_logits.hip / isg_mp_graph.hip) gathers
// x_l[src] and x_r[dst] per edge as 16-byte pieces from L2 -- 52 M lane accesses per launch at BASELINE configs[1], which is what
// holds that kernel at 210 us in every ordering tried (profiles/r02_w_edge_logits.md) -- then writes the logits, and the second
// kernel stages x_l again.  Here a workgroup owns (tile, head): the head's x_l slice of the tile's <= 64 nodes is staged ONCE
// with coalesced 512-byte row loads and serves both the logit epilogue (row gathers from LDS) and the aggregation; the logits
// never leave LDS.  x_r[dst] still comes from global memory: slots are sorted by destination, so a wave's 32 edges touch ~13
// rows.  The arithmetic is the pair's, operation for operation (same MFMA sequence, same partial-sum orders, the per-graph
// kernel's softmax and edge-id-ordered aggregation): results are bit-identical to it.
//   block  = 8 waves: wave = (32-slot half of a 64-slot chunk, one of the head's four 32-channel tiles)
//   LDS    = x_l slice 33 KB + edge panel planes 34 KB + CSR records / logits / partials 7 KB: two workgroups per CU
//   grid   = tiles x heads, the heads of a tile on one XCD (block ids 8 apart): the edge rows they all stage are L2 hits
// =====================================================================================================================
namespace isg {

constexpr int TC_ROWS = 64, TC_ECAP = 256, TC_C = 128, TC_KC = 128, TC_LDX = TC_C + 4, TC_LDA = TC_KC + 8, TC_THREADS = 256;
constexpr int TC_SMEM_BYTES = TC_ROWS * TC_LDX * 4 + 2 * 64 * TC_LDA * 2 + TC_ECAP * 16 + TC_ECAP * 4 + 4 * 64 * 4 + 64 * 4 + 68 * 4 + 2 * TC_C * 4;
static_assert(2 * TC_SMEM_BYTES <= 160 * 1024, "two workgroups per CU");

struct TcArgs {
  const float *x_l, *x_r;           // [N, H*C] rows (strides ldl / ldr), heads side by side
  const _Float16 *ep;               // edge features as scaled (hi, mid) fp16 planes in CSR SLOT order: [E][2][128] (isg_edge_planes)
  const float *ep_inv;              // [E] inverse row scales, slot order
  const _Float16 *Wf;               // lin_edge.weight [H*C, K] as fragment-major (hi, mid) planes
  const float *w_inv, *att;         // [H*C]
  const float *bias;                // [H*C], optional
  const int *rowptr, *eid, *src, *dst, *ntiles;
  const int4 *tile_info;            // {first node, nodes, first CSR slot, CSR slots} per tile (isg_tile_plan)
  const float *edge_mask, *node_mask;   // optional (NULL: the layer is not masked)
  float *out, *alpha;               // [N, H*C] (stride ldo), [E, H]
  float *rowmax;                    // [N, H], optional
  int N, E, H, K, KS, NT, ldl, ldr, ldo;
  float slope;
};

// The edge features are the same for every layer and every head of a step: their row scales and (hi, mid) fp16 planes are formed
// ONCE per batch, in CSR slot order (a tile's slots are then one contiguous 512-byte-per-slot range), instead of in every
// (layer, head, chunk) staging pass -- that conversion was ~30 % of the tile kernel's instructions.
__device__ __forceinline__ void edge_planes_row(const float *__restrict__ edge_attr, int lda, const int *__restrict__ eid, int E, int K,
                                                _Float16 *__restrict__ planes, float *__restrict__ inv_out, int slot, int c4) {
  if (slot >= E) return;
  // the plan fills eid[0 .. rowptr[N]) only (edges with an endpoint outside [0, N) are dropped): a slot beyond that holds
  // whatever the allocation held, so the id is checked before it becomes an address; such a slot gets zero planes
  const unsigned e = eid ? (unsigned)eid[slot] : (unsigned)slot;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < K && e < (unsigned)E) v = *reinterpret_cast<const float4 *>(edge_attr + (int64_t)e * lda + c4 * 4);
  const float mx = group_max<32>(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  float s, inv;
  h3_scale(mx, s, inv);
  if (c4 == 0) inv_out[slot] = inv;
  v.x *= s; v.y *= s; v.z *= s; v.w *= s;
  hf16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  hf16x4 mid = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                (_Float16)(v.w - (float)hi[3])};
  *reinterpret_cast<hf16x4 *>(planes + (int64_t)slot * 256 + c4 * 4) = hi;
  *reinterpret_cast<hf16x4 *>(planes + (int64_t)slot * 256 + 128 + c4 * 4) = mid;
}

__global__ __launch_bounds__(256) void edge_planes_kernel(const float *__restrict__ edge_attr, int lda, const int *__restrict__ eid,
                                                          int E, int K, _Float16 *__restrict__ planes, float *__restrict__ inv_out) {
  edge_planes_row(edge_attr, lda, eid, E, K, planes, inv_out, blockIdx.x * 8 + (threadIdx.x >> 5), threadIdx.x & 31);
}

// The tile plan (ONE workgroup, 22 us of barriers and LDS latency) and the edge planes (every CU, 36 us) depend on the graph plan
// only, not on each other: as one launch -- workgroup 0 plans the tiles, the others split 32 edge rows each -- the tile plan runs
// beside the planes instead of alone on the chip.
__global__ __launch_bounds__(TP_CH) void tile_plan_edge_planes_kernel(const int *__restrict__ ptr, const int *__restrict__ eptr, int B,
                                                                      int ncap, int ecap, int *__restrict__ tile_ptr,
                                                                      int *__restrict__ ntiles, int cap, int4 *__restrict__ tile_info,
                                                                      int4 *__restrict__ tile_heavy_first,
                                                                      const float *__restrict__ edge_attr, int lda,
                                                                      const int *__restrict__ eid, int E, int K,
                                                                      _Float16 *__restrict__ planes, float *__restrict__ inv_out) {
  if (blockIdx.x == 0)
    tile_plan_body(ptr, eptr, B, ncap, ecap, tile_ptr, ntiles, cap, tile_info, tile_heavy_first);
  else
    edge_planes_row(edge_attr, lda, eid, E, K, planes, inv_out, (blockIdx.x - 1) * 32 + (threadIdx.x >> 5), threadIdx.x & 31);
}

// PERSISTENT: the grid is two workgroups per CU; a workgroup keeps ONE head (its W fragments never leave its registers) and walks
// tiles t = group, group + groups, ...; while it finishes a tile (softmax, aggregation) the next tile's CSR records, row pointers,
// x_l slice and first edge rows are already in flight into registers.  A (tile, head) workgroup per launch-grid entry measured
// 317-352 us against 330 us for the pair: every workgroup paid five dependent memory round trips for ~4.6k cycles of MFMA work at
// two waves per SIMD (profiles/r03_l_tile_conv_stamps.txt).  Every wave leaves the loop at the same tile: the bound is uniform.
template <bool MASKED>
__global__ __launch_bounds__(TC_THREADS, 2) void gatv2_tile_conv_kernel(TcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];
  typedef float (*BufX)[TC_LDX];
  typedef _Float16 (*BufP)[64][TC_LDA];
  BufX sXl = reinterpret_cast<BufX>(tc_smem);
  BufP sA = reinterpret_cast<BufP>(tc_smem + TC_ROWS * TC_LDX * 4);
  int4 *s_tab = reinterpret_cast<int4 *>(tc_smem + TC_ROWS * TC_LDX * 4 + 2 * 64 * TC_LDA * 2);     // {src - r0, eid, dst - r0, mask bits}
  float *s_lg = reinterpret_cast<float *>(s_tab + TC_ECAP);
  float *s_part = s_lg + TC_ECAP;             // [4 tile-waves][64 slots]; after the chunks: the softmax weights
  float *s_inv = s_part + 4 * 64;
  int *s_rp = reinterpret_cast<int *>(s_inv + 64);
  float *s_att = reinterpret_cast<float *>(s_rp + 68), *s_winv = s_att + TC_C;     // the head's att / inverse W scales

  // workgroups 8 apart share an XCD (round-robin dispatch): the H head-workgroups of a tile group sit on one XCD, so the edge
  // rows and CSR records they all read are L2 hits for three of them
  const int bid = blockIdx.x;
  const int per_xcd = gridDim.x >> 3, j = bid >> 3;
  const int hd = j % a.H;
  const int ngrp = gridDim.x / a.H;
  int t = (bid & 7) * (per_xcd / a.H) + j / a.H;
  const int T = *a.ntiles;
  if (t >= T) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = one of the head's four 32-channel tiles, all 64 slots
  ISG_DIAG_BEGIN()
  const int fr = lane & 31, hh = lane >> 5, fk = hh * 8;
  const int hoff = hd * TC_C;
  const int srow = tid >> 5, sc4 = tid & 31;          // staging map: 32 lanes per 512-byte row, rows srow + 8 u

  // ---- W fragments of this wave's channel tile: loaded once, resident for every tile of the workgroup -----------------------
  const int nt = hd * (TC_C / 32) + tw;
  const int KS = a.KS;
  const unsigned plane_b = (unsigned)a.NT * (unsigned)KS * 1024u;
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.Wf), 0, (int)(2u * plane_b), 0x00020000);
  const unsigned wb = (unsigned)nt * (unsigned)KS * 1024u;
  hf16x8 wq[8][2];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      wq[ks][q] = hf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (ks < KS)
        wq[ks][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   wrsrc, lane * 16, (int)(wb + q * plane_b + (unsigned)ks * 1024u), 0));
    }
  if (tid < TC_C) s_att[tid] = a.att[hoff + tid];
  else s_winv[tid - TC_C] = a.w_inv[hoff + tid - TC_C];
  const float4 b4 = a.bias ? *reinterpret_cast<const float4 *>(a.bias + hoff + fr * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int cb = nt * 32 + 4 * hh;                  // this lane's channels of the tile: cb + 8 g + j
  const float slope = a.slope;

  // ---- a tile's inputs on their way into registers (issued a tile ahead), then into LDS -------------------------------------
  hf32x4 ra[8];
  // (macros, not lambdas: register arrays captured by reference end up in scratch memory)
#define TC_REQUEST_TILE(d)                           /* d = {r0, nrows, e0, ne} */                                      \
  {                                                                                                                  \
    const int r0n = (d).x, nrn = min((d).y, TC_ROWS), e0n = (d).z, nen = min((d).w, TC_ECAP);                          \
    rp_n = 0;                                                                                                        \
    if (tid <= nrn) rp_n = a.rowptr[r0n + tid] - e0n;                                                                \
    rec_n = make_int4(0, 0, 0, __float_as_int(1.f));                                                                 \
    if (tid < nen) {                                                                                                 \
      const int s_ = a.src[e0n + tid], e_ = a.eid[e0n + tid], d_ = a.dst[e0n + tid];                                 \
      rec_n.x = min(max(s_ - r0n, 0), max(nrn - 1, 0));    /* a source outside its tile is clamped into it */        \
      rec_n.y = e_;                                                                                                  \
      rec_n.z = min(max(d_ - r0n, 0), max(nrn - 1, 0));                                                              \
      sraw_n = s_;                                                                                                   \
      draw_n = d_;                                                                                                   \
    }                                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                  \
      const int row = min(r0n + min(srow + 8 * u, max(nrn - 1, 0)), a.N - 1);                                        \
      xv_n[u] = __builtin_nontemporal_load(reinterpret_cast<const hf32x4 *>(a.x_l + (int64_t)row * a.ldl + hoff + sc4 * 4)); \
    }                                                                                                                \
  }
  // chunk 0's edge planes of that tile (slot order: one contiguous range) and the mask values
#define TC_REQUEST_ROWS0(d)                                                                                          \
  {                                                                                                                  \
    const int e0n = (d).z, nen = min((d).w, TC_ECAP);                                                                \
    if (nen > 0) {                                                                                                   \
      _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                                  \
        ra[u] = *reinterpret_cast<const hf32x4 *>(a.ep + (int64_t)(e0n + min(srow + 8 * u, nen - 1)) * 256 + sc4 * 8); \
    }                                                                                                                \
    if (MASKED && tid < nen)                                                                                         \
      rec_n.w = __float_as_int(a.edge_mask ? a.edge_mask[rec_n.y] : a.node_mask[sraw_n] * a.node_mask[draw_n]);      \
  }
#define TC_STORE_TILE(d)                                                                                             \
  {                                                                                                                  \
    const int nrn = min((d).y, TC_ROWS);                                                                             \
    if (tid <= nrn) s_rp[tid] = rp_n;                                                                                \
    s_tab[tid] = rec_n;                                                                                              \
    _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                                    \
      if (srow + 8 * u < nrn) *reinterpret_cast<hf32x4 *>(&sXl[srow + 8 * u][sc4 * 4]) = xv_n[u];                    \
  }

  int4 desc = a.tile_info[t];
  {
    int4 rec_n;
    int rp_n, sraw_n = 0, draw_n = 0;
    hf32x4 xv_n[8];
    TC_REQUEST_TILE(desc)
    TC_REQUEST_ROWS0(desc)
    TC_STORE_TILE(desc)
  }
  __syncthreads();
  TC_STAMP(0)              // first tile's inputs (exposed once per workgroup)

#pragma unroll 1
  while (true) {
    const int r0 = desc.x, nrows = min(desc.y, TC_ROWS), ne = min(desc.w, TC_ECAP);
    const int t_next = t + ngrp;
    const bool has_next = t_next < T;
    int4 desc_n = make_int4(0, 0, 0, 0);
    if (has_next) desc_n = a.tile_info[t_next];

    // ---- 64-slot chunks: edge panel -> (hi, mid) planes, transposed product, logit epilogue (isg_mp_logits.hip) --------------
    const int nchunk = (ne + 63) >> 6;
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
      // this lane's two edges of the chunk (slots fr and 32 + fr): their x_r[dst] pieces are requested now, used after the k loop
      hf32x4 xr[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int dl = s_tab[min(64 * c + i * 32 + fr, ne - 1)].z;
        const float *xr_row = a.x_r + (int64_t)(r0 + dl) * a.ldr + cb;
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[i][g] = *reinterpret_cast<const hf32x4 *>(xr_row + 8 * g);
      }
      // the chunk's edge planes (requested a chunk ahead): 16-byte pieces straight into the panel image
      const int e0 = desc.z;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        *reinterpret_cast<hf32x4 *>(&sA[sc4 >> 4][srow + 8 * u][(sc4 & 15) * 8]) = ra[u];
      if (tid < 64) s_inv[tid] = a.ep_inv[e0 + min(64 * c + tid, ne - 1)];
      __syncthreads();
      TC_STAMP(2)            // x_r requests + panel staging + barrier
      if (c + 1 < nchunk) {      // the next chunk's planes: in flight under this chunk's product and epilogue
#pragma unroll
        for (int u = 0; u < 8; ++u)
          ra[u] = *reinterpret_cast<const hf32x4 *>(a.ep + (int64_t)(e0 + min(64 * (c + 1) + srow + 8 * u, ne - 1)) * 256 + sc4 * 8);
      }
      hf32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      {
        // one register set of panel fragments: a k-step's LDS reads are issued behind the previous step's six MFMAs
        hf16x8 af[2][2];          // [half][plane]
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks < KS) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int q = 0; q < 2; ++q) af[i][q] = *reinterpret_cast<const hf16x8 *>(&sA[q][i * 32 + fr][ks * 16 + fk]);
            // transposed product: W fragment = A operand (rows = channels), edge panel = B operand (columns = edges)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][1], af[i][0], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks][0], af[i][0], acc[i], 0, 0, 0);
          }
        }
      }
      ISG_DIAG_KEEP2(acc[0][0], acc[1][15])
      TC_STAMP(3)            // k loop
      // epilogue: e = acc * s_row * s_col (exact powers of two), z = (x_r[i] + x_l[j]) + e, mask, leaky, mask, z * att in 4 chains
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int prow = i * 32 + fr;
        const float sinv = s_inv[prow];
        const int4 rec = s_tab[min(64 * c + prow, ne - 1)];
        const float me = own_reg(__int_as_float(rec.w));      // a scalar: never the high dword of the record's (z, w) pair
        float part[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 xl4 = *reinterpret_cast<const float4 *>(&sXl[rec.x][tw * 32 + 4 * hh + 8 * g]);
          const float4 at4 = *reinterpret_cast<const float4 *>(&s_att[tw * 32 + 4 * hh + 8 * g]);
          const float4 wi4 = *reinterpret_cast<const float4 *>(&s_winv[tw * 32 + 4 * hh + 8 * g]);
          const float lv[4] = {xl4.x, xl4.y, xl4.z, xl4.w};
          const float atv[4] = {at4.x, at4.y, at4.z, at4.w}, wiv[4] = {wi4.x, wi4.y, wi4.z, wi4.w};
          part[g] = 0.f;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float e = (acc[i][g * 4 + jj] * sinv) * wiv[jj];
            float z = (xr[i][g][jj] + lv[jj]) + e;
            if (MASKED) z *= me;
            z = z > 0.f ? z : z * slope;
            if (MASKED) z *= me;
            part[g] = fmaf(z, atv[jj], part[g]);
          }
        }
        const float mine = (part[0] + part[1]) + (part[2] + part[3]);
        const float tot = mine + __shfl_xor(mine, 32);
        if (hh == 0) s_part[tw * 64 + prow] = tot;
      }
      __syncthreads();
      TC_STAMP(4)            // epilogue + barrier
      if (tid < 64 && 64 * c + tid < ne)     // the tile-waves' partials in a fixed order
        s_lg[64 * c + tid] = (s_part[tid] + s_part[64 + tid]) + (s_part[128 + tid] + s_part[192 + tid]);
    }
    __syncthreads();

    // ---- softmax + aggregation (isg_mp_graph.hip phase C: same operations in the same order) --------------------------------
    // C1, a thread per slot: maximum and denominator of the slot's destination segment (walked in slot order), the weight; alpha
    float *s_w = s_part;                  // the partial-logit table is free now
    if (tid < ne) {
      const int4 rc = s_tab[tid];
      const int rb = s_rp[rc.z], re = min(s_rp[rc.z + 1], ne);
      float mx = -INFINITY;
#pragma unroll 2
      for (int s = rb; s < re; ++s) mx = fmaxf(mx, s_lg[s]);
      float den = 0.f;
#pragma unroll 2
      for (int s = rb; s < re; ++s) den += __builtin_amdgcn_exp2f((s_lg[s] - mx) * 1.4426950408889634f);
      const float w = __builtin_amdgcn_exp2f((s_lg[tid] - mx) * 1.4426950408889634f) * __builtin_amdgcn_rcpf(den + 1e-16f);
      a.alpha[(int64_t)rc.y * a.H + hd] = w;
      s_w[tid] = MASKED ? mul_rn(w, __int_as_float(rc.w)) : w;
    }
    int4 rec_n;                 // the next tile's inputs: defined and consumed inside this iteration
    int rp_n, sraw_n = 0, draw_n = 0;
    hf32x4 xv_n[8];
    TC_REQUEST_TILE(desc_n)     // (unconditional: a zero descriptor requests row 0 and nothing else) the next tile's records, row pointers, x_l slice, edge ids: in flight under C2
    __syncthreads();
    TC_STAMP(6)              // weights
    // C2, a half-wave per destination node, lane = float4 column: x_l rows from LDS in edge-id order, one fma per term
#pragma unroll 1
    for (int k = 2 * tw + hh; k < nrows; k += 8) {
      const int rb = s_rp[k], re = min(s_rp[k + 1], ne);
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      // One in-edge per trip, its weight in a register of its own (own_reg).  Round 4's `#pragma unroll 2` form let the second
      // fma of a pair take its weight from the HIGH dword of a ds_read2_b32 result (v_pk_fma_f32 .. op_sel:[0,1,0]) and one launch
      // in ~15 left o.x or o.z of lanes 16-31 of a half-wave without that in-edge's term.  tools/flake/ (DESIGN.md 16.1): 48-417
      // wrong launches of 1600 in every variant with that operand form -- two accumulators, no loads in flight, 16 idle cycles
      // behind the LDS reads alike -- and 0 of 1600 in the variants without it; tools/scan_pk_cross.py keeps the form out of the
      // whole library.
#pragma unroll 1
      for (int s = rb; s < re; ++s) {
        const float wm = own_reg(s_w[s]);
        const float4 u4 = *reinterpret_cast<const float4 *>(&sXl[s_tab[s].x][fr * 4]);
        o.x = fmaf(u4.x, wm, o.x);
        o.y = fmaf(u4.y, wm, o.y);
        o.z = fmaf(u4.z, wm, o.z);
        o.w = fmaf(u4.w, wm, o.w);
      }
      if (a.bias) { o.x += b4.x; o.y += b4.y; o.z += b4.z; o.w += b4.w; }
      hf32x4 o4 = {o.x, o.y, o.z, o.w};
      __builtin_nontemporal_store(o4, reinterpret_cast<hf32x4 *>(a.out + (int64_t)(r0 + k) * a.ldo + hoff + fr * 4));
      if (a.rowmax) {
        const float rmx = group_max<32>(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        if (fr == 0) a.rowmax[(int64_t)(r0 + k) * a.H + hd] = rmx;
      }
    }
    TC_STAMP(7)              // aggregation, stores
    if (!has_next) break;
    TC_REQUEST_ROWS0(desc_n)
    __syncthreads();         // every wave is done with this tile's LDS image
    TC_STORE_TILE(desc_n)
    __syncthreads();
    TC_STAMP(1)              // hand-over to the next tile
    desc = desc_n;
    t = t_next;
  }
#undef TC_REQUEST_TILE
#undef TC_REQUEST_ROWS0
#undef TC_STORE_TILE
  ISG_DIAG_DUMP(g_dt_stamps, bid * 4 + tw, 12, )
}

}  // namespace isg

using namespace isg;

// MaskingGATv2Conv.message + aggregate with lin_edge inside, on graph-aligned tiles (isg_tile_plan with node_cap = 64 and
// edge_cap = 256, tile_info requested) and on the edge features' planes (isg_edge_planes): see the kernel's header.  ISG_EUNSUPPORTED unless C == 128, K <= 128, K % 4 == 0.
extern "C" int isg_gatv2_tile_conv(const float *x_l, int32_t ldl, const float *x_r, int32_t ldr, const uint16_t *edge_planes,
                                   const float *edge_inv_scale, const uint16_t *w_frag, const float *w_inv_scale, const float *att,
                                   const float *bias, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                                   const int32_t *dst, const int32_t *tile_info, const int32_t *ntiles, int64_t max_tiles,
                                   const float *node_mask, const float *edge_mask, float *out, int32_t ldo, float *alpha,
                                   float *rowmax, int64_t N, int64_t E, int32_t H, int32_t C, int32_t K, float negative_slope,
                                   void *stream) {
  if (N < 0 || E < 0 || H <= 0 || C <= 0 || K <= 0 || max_tiles < 0 || ldl < H * C || ldr < H * C || ldo < H * C)
    return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (C != TC_C || K > TC_KC || (K & 3) != 0 || (ldl & 3) != 0 || (ldr & 3) != 0 || (ldo & 3) != 0 || H > 64 ||
      mis(x_l) || mis(x_r) || mis(edge_planes) || mis(att) || mis(w_inv_scale) || mis(out) || (bias && mis(bias)) || mis(tile_info) ||
      N >= (1ll << 31) || E >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  if (N == 0 || max_tiles == 0) return ISG_OK;
  if (!x_l || !x_r || (E > 0 && (!edge_planes || !edge_inv_scale || !eid || !src || !dst || !alpha)) || !w_frag || !w_inv_scale || !att || !rowptr ||
      !tile_info || !ntiles || !out)
    return ISG_EINVAL;
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  TcArgs a = {
      .x_l = x_l, .x_r = x_r, .ep = reinterpret_cast<const _Float16 *>(edge_planes), .ep_inv = edge_inv_scale,
      .Wf = reinterpret_cast<const _Float16 *>(w_frag), .w_inv = w_inv_scale, .att = att, .bias = bias, .rowptr = rowptr, .eid = eid,
      .src = src, .dst = dst, .ntiles = ntiles, .tile_info = reinterpret_cast<const int4 *>(tile_info), .edge_mask = edge_mask,
      .node_mask = node_mask, .out = out, .alpha = alpha, .rowmax = rowmax, .N = (int)N, .E = (int)E, .H = H, .K = K,
      .KS = (K + 15) / 16, .NT = H * C / 32, .ldl = ldl, .ldr = ldr, .ldo = ldo, .slope = negative_slope};
  if (!a.x_l || !a.x_r || (a.E > 0 && (!a.ep || !a.ep_inv || !a.eid || !a.src || !a.dst || !a.alpha)) || !a.Wf || !a.w_inv || !a.att ||
      !a.rowptr || !a.tile_info || !a.ntiles || !a.out)
    return ISG_EINVAL;                         // the struct the kernel dereferences, not the parameters it was filled from
  // two workgroups per CU (LDS), 256 CUs: 8 XCDs x (groups per XCD) x H head-workgroups; fewer groups when there are few tiles
  const int gpx_max = device_cus();
  int gpx = (2 * gpx_max / 8) / H;                       // groups per XCD
  if (gpx < 1) gpx = 1;
  const long long need = (max_tiles + 7) / 8;            // groups per XCD that would each get one tile
  if (gpx > need) gpx = (int)need;
  const unsigned grid = 8u * (unsigned)H * (unsigned)gpx;
  hipStream_t st = as_stream(stream);
  if (node_mask || edge_mask) {
    if (!dyn_lds_ok<&gatv2_tile_conv_kernel<true>>(TC_SMEM_BYTES)) return ISG_EUNSUPPORTED;
    gatv2_tile_conv_kernel<true><<<grid, TC_THREADS, TC_SMEM_BYTES, st>>>(a);
  } else {
    if (!dyn_lds_ok<&gatv2_tile_conv_kernel<false>>(TC_SMEM_BYTES)) return ISG_EUNSUPPORTED;
    gatv2_tile_conv_kernel<false><<<grid, TC_THREADS, TC_SMEM_BYTES, st>>>(a);
  }
  return check_launch();
}

// Edge features -> per-row power-of-two scale and (hi, mid) fp16 planes in CSR SLOT order, once per batch: planes
// uint16 [E][2][128] (row e: the 128 hi values, then the 128 mid values; columns beyond K are zero), inv_scale fp32 [E].
// The operand of isg_gatv2_tile_conv for every layer and head.  ISG_EUNSUPPORTED unless K <= 128, K % 4 == 0.
extern "C" int isg_edge_planes(const float *edge_attr, int32_t lda, const int32_t *eid, int64_t E, int32_t K, uint16_t *planes,
                               float *inv_scale, void *stream) {
  if (E < 0 || K <= 0 || lda < K) return ISG_EINVAL;
  if (K > 128 || (K & 3) != 0 || (lda & 3) != 0 || (reinterpret_cast<uintptr_t>(edge_attr) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(planes) & 15) != 0 || E >= (1ll << 31) - 8)
    return ISG_EUNSUPPORTED;
  if (E == 0) return ISG_OK;
  if (!edge_attr || !planes || !inv_scale) return ISG_EINVAL;
  edge_planes_kernel<<<(unsigned)((E + 7) / 8), 256, 0, as_stream(stream)>>>(edge_attr, lda, eid, (int)E, K,
                                                                             reinterpret_cast<_Float16 *>(planes), inv_scale);
  return check_launch();
}

// isg_tile_plan + isg_edge_planes as ONE launch (see tile_plan_edge_planes_kernel): same operands, same results.
extern "C" int isg_tile_plan_edge_planes(const int32_t *ptr, const int32_t *eptr, int64_t B, int32_t node_cap, int32_t edge_cap,
                                         int32_t *tile_ptr, int32_t *ntiles, int32_t *tile_info, int64_t capacity,
                                         int32_t *tile_info_heavy_first, const float *edge_attr, int32_t lda, const int32_t *eid,
                                         int64_t E, int32_t K, uint16_t *planes, float *inv_scale, void *stream) {
  if (B < 0 || node_cap <= 0 || capacity < 0 || !tile_ptr || !ntiles || (B > 0 && !ptr) || E < 0 || K <= 0 || lda < K ||
      (tile_info_heavy_first && (!tile_info || (reinterpret_cast<uintptr_t>(tile_info_heavy_first) & 15) != 0)))
    return ISG_EINVAL;
  if (eptr && edge_cap <= 0) return ISG_EINVAL;
  if (B >= (1ll << 31) || capacity >= (1ll << 31) || K > 128 || (K & 3) != 0 || (lda & 3) != 0 ||
      (reinterpret_cast<uintptr_t>(edge_attr) & 15) != 0 || (reinterpret_cast<uintptr_t>(planes) & 15) != 0 || E >= (1ll << 31) - 1024)
    return ISG_EUNSUPPORTED;
  if (tile_info && (reinterpret_cast<uintptr_t>(tile_info) & 15) != 0) return ISG_EINVAL;
  if (E > 0 && (!edge_attr || !planes || !inv_scale)) return ISG_EINVAL;
  tile_plan_edge_planes_kernel<<<(unsigned)(1 + (E + 31) / 32), TP_CH, 0, as_stream(stream)>>>(
      ptr, eptr, (int)B, node_cap, edge_cap, tile_ptr, ntiles, (int)capacity, reinterpret_cast<int4 *>(tile_info),
      reinterpret_cast<int4 *>(tile_info_heavy_first), edge_attr, lda, eid, (int)E, K, reinterpret_cast<_Float16 *>(planes), inv_scale);
  return check_launch();
}
