// fp32 Linears for K >= 256 on the fp16 matrix cores: D[M,N] = act(A[M,K] . W[N,K]^T + bias), both operands handed over
// PRE-SPLIT ("planes32"), the whole operand path LDS-DMA.
//
// Numerics are isg_gemm_f16x3.hip's: every row of A and of W is scaled by its own power of two into fp16's range and split
// into two fp16 planes (hi + mid = the scaled value to 2^-24), hi*mid + mid*hi + hi*hi accumulated in fp32, the two scales
// taken back out in the epilogue (exact).  What is different is who splits and how the bytes travel:
//   * planes32 layout: row r, k-tile kt (32 columns) = ONE 128-byte line [hi 32 | mid 32]; rows are KT lines long.  The
//     producer of an activation writes it in this form (isg_split_planes32, or this kernel's own epilogue), so the GEMM
//     converts nothing and both operands go global -> LDS by `global_load_lds` (16 bytes per lane, 1 KB per wave
//     instruction) with no register staging at all.
//   * a row's scale need not come from the row's true maximum: ANY power of two that bounds the row keeps the split exact
//     to 2^-24 of the bound (the fp16 subnormal spacing is absolute), so a GEMM can emit its result as planes with the
//     scale 2^14 * inv_a[m] * max_n ||W_n||_1 + max |b| known before the first product -- the same for every column tile.
//   * tile 256 x 256 x 32, 8 waves as 2 x 4 (128 x 64 per wave = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16, 96
//     MFMAs per wave and k-tile for 24 fragment reads), two 64 KB LDS buffers, one workgroup per CU.  A k-tile is four
//     PHASES (one quadrant of the wave's accumulators each); the two wave groups (rows 0-127 / 128-255: one wave of each
//     per SIMD) run half a phase apart, so that on every SIMD one wave issues MFMAs while its partner reads fragments
//     and issues the next DMA requests (MI355X_MICROARCH.md, "Two waves per SIMD").
//   * every phase stages one quarter of a future k-tile (2 DMA requests per thread); four quarters are in flight at any
//     time, retired by a counted `s_waitcnt vmcnt(6)` a phase before their first read.  Image rows are cut so that a
//     quarter is what ONE phase reads: A-lo / A-hi = the first / second 64 rows of BOTH wave groups, B-lo / B-hi = the
//     first / second 32 columns of all four wave columns.  16-byte pieces are XOR-swizzled on the DMA's SOURCE address
//     (piece ^ ((row >> 1) & 7)) so that the 16 lanes of a `ds_read_b128` group hit 16 different bank slots.
//   * the product is formed transposed (W fragment = the MFMA's A operand): a lane ends up with one row and four
//     consecutive columns, i.e. 16-byte stores straight from the accumulators.
#include "isg_f16x3.hpp"

#include "isg_diag.hpp"

#include <stdlib.h>

// -DISG_DIAG (tools/stamp_h3p.py builds its own library): per-wave s_memtime totals of the persistent kernel's segments
ISG_DIAG_BUFFER(g_p3_stamps)            // [workgroups * 8 waves][16] int64
#define Q3_ST(i) ISG_DIAG_ADD(i)

namespace isg {

typedef __attribute__((address_space(3))) void p3_lds_t;
typedef __attribute__((address_space(1))) void p3_glb_t;
typedef unsigned int p3_u32x4 __attribute__((ext_vector_type(4)));
typedef int p3_i32x4 __attribute__((ext_vector_type(4)));

constexpr int P3_T = 256, P3_THREADS = 512;
constexpr int P3_SLOT = 128 * 128;          // one quarter image: 128 rows x (hi 32 | mid 32) fp16
constexpr int P3_BUF = 4 * P3_SLOT;         // A-lo, A-hi, B-lo, B-hi of one k-tile
constexpr int P3_SMEM = 2 * P3_BUF;         // 128 KB
constexpr int P3_ALO = 0, P3_AHI = 1, P3_BLO = 2, P3_BHI = 3;

// planes32 of fp32 rows; the row's largest magnitude decides its scale
template <int LPR, int PMAX>
__global__ __launch_bounds__(256) void split_planes32_kernel(const float *__restrict__ a, int M, int K, int lda,
                                                             _Float16 *__restrict__ planes, float *__restrict__ inv_out) {
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR, l = threadIdx.x % LPR;
  if (row >= M) return;
  const int KT = (K + 31) >> 5, nc = K >> 2;
  const float4 *r4 = reinterpret_cast<const float4 *>(a + (int64_t)row * lda);
  planes32_row<LPR, PMAX>(l, nc, KT, planes + (int64_t)row * KT * 64, inv_out + row, [&](int c) { return r4[c]; },
                          [](int, float4) {});
}

// gelu(x * instr[batch]) (ISubGVQA/models/mgat_v2_conv.py:156-157) written as the planes32 operand of the lin_l | lin_r
// projection, and as fp32 rows where a masked layer's node gate reads them: the instruction gate and the split as ONE pass over
// the layer input.
template <int LPR, int PMAX>
__global__ __launch_bounds__(256) void instr_gate_planes32_kernel(const float *__restrict__ x, const float *__restrict__ instr,
                                                                  const long long *__restrict__ batch, int N, int C,
                                                                  float *__restrict__ rows, _Float16 *__restrict__ planes,
                                                                  float *__restrict__ inv_out) {
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR, l = threadIdx.x % LPR;
  if (row >= N) return;
  const int KT = (C + 31) >> 5, nc = C >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x + (int64_t)row * C);
  const float4 *i4 = reinterpret_cast<const float4 *>(instr + batch[row] * (int64_t)C);
  float4 *o4 = rows ? reinterpret_cast<float4 *>(rows + (int64_t)row * C) : nullptr;
  planes32_row<LPR, PMAX>(l, nc, KT, planes + (int64_t)row * KT * 64, inv_out + row,
                          [&](int c) {
                            const float4 a = x4[c], b = i4[c];
                            const isg_f32x2 g0 = gelu_exact2(isg_f32x2{a.x * b.x, a.y * b.y}), g1 = gelu_exact2(isg_f32x2{a.z * b.z, a.w * b.w});
                            return make_float4(g0.x, g0.y, g1.x, g1.y);
                          },
                          [&](int c, float4 t) { if (o4) o4[c] = t; });
}

// launch of a row -> planes32 kernel: sixteen lanes per row with the row in registers up to 128 float4, else a wave per row
#define ISG_PLANES32_ROWS_LAUNCH(kern, rows_, nc_, st_, ...)                                                       \
  do {                                                                                                             \
    const int pad_ = ((nc_) + 7) / 8 * 8;                                                                          \
    if (pad_ <= 80) kern<16, 5><<<(unsigned)(((rows_) + 15) / 16), 256, 0, st_>>>(__VA_ARGS__);                   \
    else if (pad_ <= 128) kern<16, 8><<<(unsigned)(((rows_) + 15) / 16), 256, 0, st_>>>(__VA_ARGS__);             \
    else if (pad_ <= 512) kern<64, 8><<<(unsigned)(((rows_) + 3) / 4), 256, 0, st_>>>(__VA_ARGS__);               \
    else kern<64, 0><<<(unsigned)(((rows_) + 3) / 4), 256, 0, st_>>>(__VA_ARGS__);                                \
  } while (0)

struct P3Args {
  const _Float16 *A, *W;          // planes32 [M][KT][64], [N][KT][64]
  const float *a_inv, *w_inv;     // [M], [N]
  const float *bias;              // [N] or NULL
  float *D;                       // fp32 [M, ldd], or NULL with the planes output below
  _Float16 *Dp;                   // planes32 [M][N / 32][64]
  float *d_inv;                   // [M]
  const float *d_bound;           // {2^14 * max_n ||W_n||_1, max |b|}: |D[m, :]| < inv_a[m] * d_bound[0] + d_bound[1]
  int M, N, KT, ldd, tiles_n, nt_store;
  const float *a_inv0;            // SEGMENTED A (persistent form): the row scales of k-tiles [0, kseg); a_inv then holds those of
  int kseg;                       // [kseg, KT).  NULL / 0: one scale per row
};

template <int ACT, bool PLANES_OUT>
__global__ __launch_bounds__(P3_THREADS, 2) void linear_h3p_kernel(const P3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p3_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int m0, n0;
  {                                        // XCD-aware order: the column tiles of a row panel run back to back on ONE XCD
    const int L = blockIdx.x, slot = L >> 3;
    m0 = ((L & 7) + 8 * (slot / a.tiles_n)) * P3_T;
    n0 = (slot % a.tiles_n) * P3_T;
    if (m0 >= a.M) return;
  }
  const int nk = a.KT;
  const unsigned row_b = (unsigned)a.KT * 128u;          // bytes per operand row

  // ---- DMA sources: quarter image q, request u -> this lane's 16 bytes --------------------------------------------
  unsigned soff[4][2];
  {
    const int piece = (lane & 7) ^ ((lane >> 4) | ((wave & 1) << 2));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = (u * 8 + wave) * 8 + (lane >> 3);            // image row 0..127
      const int alo = r < 64 ? r : r + 64, ahi = alo + 64;       // tile rows of A-lo / A-hi
      const int blo = (r >> 5) * 64 + (r & 31), bhi = blo + 32;  // tile columns of B-lo / B-hi
      soff[P3_ALO][u] = (unsigned)min(m0 + alo, a.M - 1) * row_b + piece * 16;
      soff[P3_AHI][u] = (unsigned)min(m0 + ahi, a.M - 1) * row_b + piece * 16;
      soff[P3_BLO][u] = (unsigned)min(n0 + blo, a.N - 1) * row_b + piece * 16;
      soff[P3_BHI][u] = (unsigned)min(n0 + bhi, a.N - 1) * row_b + piece * 16;
    }
  }
  const unsigned char *Ab = reinterpret_cast<const unsigned char *>(a.A);
  const unsigned char *Wb = reinterpret_cast<const unsigned char *>(a.W);
#define P3_STAGE(q, kt, buf)                                                                                       \
  {                                                                                                                \
    const unsigned ko = (unsigned)min((kt), nk - 1) * 128u;                                                        \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                  \
      __builtin_amdgcn_global_load_lds((p3_glb_t *)(((q) < 2 ? Ab : Wb) + soff[q][u] + ko),                        \
                                       (p3_lds_t *)(p3_smem + (buf) * P3_BUF + (q) * P3_SLOT + (u * 8 + wave) * 1024), \
                                       16, 0, 0);                                                                  \
  }

  // ---- fragment addresses (bytes inside a quarter image) ------------------------------------------------------------
  const int pz = ((lane >> 4) ^ ((lane >> 1) & 7)) * 16;
  const int a_rd = (wr * 64 + (lane & 15)) * 128 + pz;           // + i * 2048; mid plane: ^ 64
  const int b_rd = (wc * 32 + (lane & 15)) * 128 + pz;           // + j * 2048

  hf32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = hf32x4{0.f, 0.f, 0.f, 0.f};
  hf16x8 fa[4][2], fbl[2][2], fbh[2][2];

#define P3_READ_A(q, buf)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
    fa[i][0] = *reinterpret_cast<const hf16x8 *>(p3_smem + (buf) * P3_BUF + (q) * P3_SLOT + i * 2048 + a_rd);      \
    fa[i][1] = *reinterpret_cast<const hf16x8 *>(p3_smem + (buf) * P3_BUF + (q) * P3_SLOT + i * 2048 + (a_rd ^ 64)); \
  }
#define P3_READ_B(FB, q, buf)                                                                                      \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                  \
    FB[j][0] = *reinterpret_cast<const hf16x8 *>(p3_smem + (buf) * P3_BUF + (q) * P3_SLOT + j * 2048 + b_rd);      \
    FB[j][1] = *reinterpret_cast<const hf16x8 *>(p3_smem + (buf) * P3_BUF + (q) * P3_SLOT + j * 2048 + (b_rd ^ 64)); \
  }
  // one quadrant: 4 row tiles x 2 column tiles x 3 products, small terms first
#define P3_MMA(FB, ah, bh)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                    \
    hf32x4 c = acc[(ah) * 4 + i][(bh) * 2 + j];                                                                    \
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(FB[j][1], fa[i][0], c, 0, 0, 0);                                    \
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(FB[j][0], fa[i][1], c, 0, 0, 0);                                    \
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(FB[j][0], fa[i][0], c, 0, 0, 0);                                    \
    acc[(ah) * 4 + i][(bh) * 2 + j] = c;                                                                           \
  }
  // end of a load segment: the three youngest quarters stay in flight; then the hand-over barrier
#define P3_L_END                                                                                                   \
  ISG_WAIT(0x0F76);   /* vmcnt(6) */                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  ISG_BARRIER();                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define P3_M(FB, ah, bh)                                                                                           \
  ISG_WAIT(0xC07F);   /* lgkmcnt(0) */                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_setprio(1);                                                                                   \
  P3_MMA(FB, ah, bh)                                                                                               \
  __builtin_amdgcn_s_setprio(0);                                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  ISG_BARRIER();                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define P3_TILE(t, buf)                                                                                            \
  {                                                                                                                \
    P3_READ_A(P3_ALO, buf)                                                                                         \
    P3_READ_B(fbl, P3_BLO, buf)                                                                                    \
    P3_STAGE(P3_BLO, (t) + 1, (buf) ^ 1)                                                                           \
    P3_L_END                                                                                                       \
    P3_M(fbl, 0, 0)                                                                                                \
    P3_READ_B(fbh, P3_BHI, buf)                                                                                    \
    P3_STAGE(P3_BHI, (t) + 1, (buf) ^ 1)                                                                           \
    P3_L_END                                                                                                       \
    P3_M(fbh, 0, 1)                                                                                                \
    P3_READ_A(P3_AHI, buf)                                                                                         \
    P3_STAGE(P3_AHI, (t) + 1, (buf) ^ 1)                                                                           \
    P3_L_END                                                                                                       \
    P3_M(fbh, 1, 1)                                                                                                \
    P3_STAGE(P3_ALO, (t) + 2, buf)                                                                                 \
    P3_L_END                                                                                                       \
    P3_M(fbl, 1, 0)                                                                                                \
  }

  // prologue: k-tile 0 and the first quarter of k-tile 1
  P3_STAGE(P3_ALO, 0, 0)
  P3_STAGE(P3_BLO, 0, 0)
  P3_STAGE(P3_BHI, 0, 0)
  P3_STAGE(P3_AHI, 0, 0)
  P3_STAGE(P3_ALO, 1, 1)
  P3_L_END
  if (wr == 1) {                         // the second wave group runs one barrier behind the first
    ISG_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int t = 0; t < nk; t += 2) {
    P3_TILE(t, 0)
    if (t + 1 < nk) P3_TILE(t + 1, 1)
  }
  if (wr == 0) {
    ISG_BARRIER();
    __builtin_amdgcn_sched_barrier(0);
  }
  ISG_WAIT(0x0F70);    // the requests past the last k-tile have landed: nothing in flight at the end
#undef P3_TILE
#undef P3_M
#undef P3_L_END
#undef P3_MMA
#undef P3_READ_B
#undef P3_READ_A
#undef P3_STAGE

  // ---- epilogue: scales out, + bias, activation; 16-byte stores straight from the accumulators -------------------------
  const int cq = 4 * (lane >> 4);
  float ia[8];
  int rows[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    rows[mi] = m0 + wr * 128 + (mi >> 2) * 64 + (mi & 3) * 16 + (lane & 15);
    ia[mi] = a.a_inv[min(rows[mi], a.M - 1)];
  }
  float so[8];
  if constexpr (PLANES_OUT) {
    const float bm = a.d_bound[0], ba = a.d_bound[1];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      float inv;
      h3_scale(ia[mi] * bm + ba, so[mi], inv);
      if (n0 == 0 && wc == 0 && lane < 16 && rows[mi] < a.M) a.d_inv[rows[mi]] = inv;
    }
  }
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) {
    const int col = n0 + wc * 64 + (nj >> 1) * 32 + (nj & 1) * 16 + cq;
    const int colc = min(col, a.N - 4);
    const float4 iw = *reinterpret_cast<const float4 *>(a.w_inv + colc);
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) b4 = *reinterpret_cast<const float4 *>(a.bias + colc);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const hf32x4 c = acc[mi][nj];
      float4 v;
      v.x = (c[0] * ia[mi]) * iw.x + b4.x;
      v.y = (c[1] * ia[mi]) * iw.y + b4.y;
      v.z = (c[2] * ia[mi]) * iw.z + b4.z;
      v.w = (c[3] * ia[mi]) * iw.w + b4.w;
      if (ACT == 1) {
        const isg_f32x2 g0 = gelu_exact2(isg_f32x2{v.x, v.y}), g1 = gelu_exact2(isg_f32x2{v.z, v.w});
        v.x = g0.x; v.y = g0.y; v.z = g1.x; v.w = g1.y;
      }
      if (ACT == 2) {     // ReLU (a NaN stays a NaN, as in torch)
        v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
      }
      if (rows[mi] < a.M && col < a.N) {
        if constexpr (PLANES_OUT) {
          const float s = so[mi];
          v.x *= s; v.y *= s; v.z *= s; v.w *= s;
          const hf16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
          const hf16x4 mid = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]),
                              (_Float16)(v.z - (float)hi[2]), (_Float16)(v.w - (float)hi[3])};
          _Float16 *d = a.Dp + (int64_t)rows[mi] * (a.N * 2) + (col >> 5) * 64 + (col & 31);
          *reinterpret_cast<hf16x4 *>(d) = hi;
          *reinterpret_cast<hf16x4 *>(d + 32) = mid;
        } else {
          typedef float p3_f32x4 __attribute__((ext_vector_type(4)));
          p3_f32x4 w4 = {v.x, v.y, v.z, v.w};
          p3_f32x4 *dst = reinterpret_cast<p3_f32x4 *>(a.D + (int64_t)rows[mi] * a.ldd + col);
          if (a.nt_store) __builtin_nontemporal_store(w4, dst);
          else *dst = w4;
        }
      }
    }
  }
}


// =====================================================================================================================
// PERSISTENT form (the default).  The 256 x 256 kernel above spends 16 us of a 48 us tile (K = 512) draining its result -- every
// CU bursts 256 KB at the same moment, the chip's HBM write rate is all they get -- and ~5 us filling its pipeline, with the
// matrix pipe idle in both (profiles/r04_b_h3p_ablation.txt); its main loop alone runs at 1.6 PF/s of fp16 products.  Here:
//   * one workgroup per CU walks tiles L = b, b + G, ...; tile 256 x 128 x 32, 8 waves as 4 x 2 (64 x 64 per wave), so a
//     wave holds TWO accumulator sets: `acc` of the tile in flight and `res` of the tile before it, whose epilogue (scales,
//     bias, activation, one 16-byte store per lane) is issued one accumulator tile per phase through the first eight k-tiles
//     of the next tile: the result leaves the chip as a steady stream beside the MFMAs instead of a burst between them;
//   * the DMA stream never stops at a tile boundary: a ring of three 48 KB k-tile buffers is filled two k-tiles ahead of the
//     reads, across tiles (a k-tile = A-lo, A-hi (the first / second 32 rows of the four wave rows) and B: three 16 KB
//     images, three requests per thread and phase); a tile's row / column scales and bias arrive the same way, one 4-byte
//     request per wave, a tile ahead;
//   * everything else (planes32 operands, source-side swizzle, ping-pong wave groups, counted vmcnt, transposed product)
//     is the kernel above.  A k-tile is two phases of 24 MFMAs.
// =====================================================================================================================
constexpr int Q3_BM = 256, Q3_BN = 128;
constexpr int Q3_BUF = 3 * P3_SLOT;            // A-lo, A-hi, B
constexpr int Q3_RING = 3 * Q3_BUF;            // 147456
constexpr int Q3_PAR = 4096;                   // per tile: a_inv[256] | w_inv[128] | bias[128] | segmented A only: a_inv0[256] | max(a_inv0, a_inv)[256]
constexpr int Q3_SMEM = Q3_RING + 3 * Q3_PAR + 256;  // 160000 (the last 256 bytes: a scratch line)
constexpr int Q3_HEAD = 8;                     // fewest k-tiles per tile: 16 accumulator tiles leave two per k-tile (K < 512), else one per k-tile over 16

struct Q3Args {
  P3Args p;
  int tiles_m, total_l;                        // row tiles; tiles_n * roundup8(tiles_m)
};

// SEG: the A operand's k-tiles [0, kseg) and [kseg, KT) carry different row scales (a_inv0 / a_inv: the flat message-passing
// kernel writes half rows, each under its own scale).  The accumulators are in units of the first scale until k-tile kseg, where
// they are multiplied by a_inv0 / a_inv -- a power of two, exact -- and from there in units of the second, which the epilogue
// undoes as always.  kseg >= HEAD: the k-tiles that carry pieces are unrolled and know nothing of it.
template <int ACT, bool PLANES_OUT, int HEAD, bool SEG = false>
__global__ __launch_bounds__(P3_THREADS, 2) void linear_h3q_kernel(const Q3Args qa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p3_smem[];
  const P3Args &a = qa.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;       // wn = the wave group (ping-pong half)
  const int nk = a.KT, G = gridDim.x;
  const unsigned row_b = (unsigned)a.KT * 128u;
  const unsigned char *Ab = reinterpret_cast<const unsigned char *>(a.A);
  const unsigned char *Wb = reinterpret_cast<const unsigned char *>(a.W);

  // tile L of the XCD-aware order -> origin; false: a padding entry of the order (row tile beyond M)
  auto origin = [&](int L, int &m0, int &n0) -> bool {
    const int slot = L >> 3;
    m0 = ((L & 7) + 8 * (slot / a.tiles_n)) * Q3_BM;
    n0 = (slot % a.tiles_n) * Q3_BN;
    return m0 < a.M;
  };
  auto next_tile = [&](int L, int &m0, int &n0) -> int {      // first valid tile of this workgroup after L, or -1
    for (L += G; L < qa.total_l; L += G)
      if (origin(L, m0, n0)) return L;
    return -1;
  };
  int cL, cm0 = 0, cn0 = 0;                                    // the tile being accumulated (its origin)
  cL = blockIdx.x;
  if (cL >= qa.total_l) return;
  if (!origin(cL, cm0, cn0)) {
    cL = next_tile(cL, cm0, cn0);
    if (cL < 0) return;
  }

  // ---- the stager: stream position (tile, k-tile) two k-tiles ahead of the reads ----------------------------------------
  const int piece16 = ((lane & 7) ^ ((lane >> 4) | ((wave & 1) << 2))) * 16;
  const int img_r0 = wave * 8 + (lane >> 3), img_r1 = img_r0 + 64;             // image rows of requests u = 0, 1
  const int alo0 = (img_r0 >> 5) * 64 + (img_r0 & 31), alo1 = (img_r1 >> 5) * 64 + (img_r1 & 31);   // tile rows of A-lo
  unsigned so_alo[2], so_ahi[2], so_b[2];
  int sL = cL, sm0 = cm0, sn0 = cn0, skt = 0;
  bool s_live = true;                                          // false: past this workgroup's last tile (requests repeat, unread)
#define Q3_SOFF()                                                                                                  \
  {                                                                                                                \
    so_alo[0] = (unsigned)min(sm0 + alo0, a.M - 1) * row_b + piece16;                                              \
    so_alo[1] = (unsigned)min(sm0 + alo1, a.M - 1) * row_b + piece16;                                              \
    so_ahi[0] = (unsigned)min(sm0 + alo0 + 32, a.M - 1) * row_b + piece16;                                         \
    so_ahi[1] = (unsigned)min(sm0 + alo1 + 32, a.M - 1) * row_b + piece16;                                         \
    so_b[0] = (unsigned)min(sn0 + img_r0, a.N - 1) * row_b + piece16;                                              \
    so_b[1] = (unsigned)min(sn0 + img_r1, a.N - 1) * row_b + piece16;                                              \
  }
  Q3_SOFF()
  int sb = 0;                                                  // ring buffer (byte offset) the stager fills
#ifndef ISG_Q3_AUX_A
#define ISG_Q3_AUX_A 0        // cache policy bits of the A-operand requests (2 = nt): A/B in tools/time_h3p.py builds
#endif
#ifndef ISG_Q3_AUX_B
#define ISG_Q3_AUX_B 0
#endif
#define Q3_DMA(base, off, slice, u)                                                                                \
  __builtin_amdgcn_global_load_lds((p3_glb_t *)((base) + (size_t)skt * 128u + (off)),                              \
                                   (p3_lds_t *)(p3_smem + sb + (slice) * P3_SLOT + ((u) * 8 + wave) * 1024), 16, 0,  \
                                   (slice) == 2 ? ISG_Q3_AUX_B : ISG_Q3_AUX_A);
#define Q3_STAGE_P0() { Q3_DMA(Wb, so_b[0], 2, 0) Q3_DMA(Wb, so_b[1], 2, 1) Q3_DMA(Ab, so_alo[0], 0, 0) Q3_DMA(Ab, so_alo[1], 0, 1) }
#define Q3_STAGE_P1()                                                                                              \
  {                                                                                                                \
    Q3_DMA(Ab, so_ahi[0], 1, 0) Q3_DMA(Ab, so_ahi[1], 1, 1)                                                        \
    sb = sb == 2 * Q3_BUF ? 0 : sb + Q3_BUF;                                                                       \
    if (s_live && ++skt == nk) {                                                                                   \
      int m1, n1;                                                                                                  \
      const int L1 = next_tile(sL, m1, n1);                                                                        \
      if (L1 < 0) { s_live = false; skt = nk - 1; }                                                                \
      else { sL = L1; sm0 = m1; sn0 = n1; skt = 0; Q3_SOFF() }                                                     \
    }                                                                                                              \
  }
  // a tile's scales and bias: ONE 4-byte request per wave (waves 0-3: a_inv of 64 rows each, 4 / 5: w_inv, 6 / 7: bias)
#define Q3_PARAMS(pm0, pn0, region)                                                                                \
  {                                                                                                                \
    const float *src = wave < 4 ? a.a_inv + min((pm0) + wave * 64 + lane, a.M - 1)                                 \
                       : (wave < 6 || !a.bias) ? a.w_inv + min((pn0) + (wave & 1) * 64 + lane, a.N - 1)            \
                                               : a.bias + min((pn0) + (wave & 1) * 64 + lane, a.N - 1);            \
    const int dst = wave < 4 ? wave * 256 : wave < 6 ? 1024 + (wave & 1) * 256 : 1536 + (wave & 1) * 256;          \
    __builtin_amdgcn_global_load_lds((p3_glb_t *)src, (p3_lds_t *)(p3_smem + Q3_RING + (region) * Q3_PAR + dst), 4, 0, 0); \
    if constexpr (SEG) {      /* every wave a second request (the counted waits are per wave and uniform): a_inv0 by waves 0-3, */ \
      const float *src0 = a.a_inv0 + min((pm0) + (wave & 3) * 64 + lane, a.M - 1);      /* again, identically, by waves 4-7 */ \
      __builtin_amdgcn_global_load_lds((p3_glb_t *)src0,                                                           \
                                       (p3_lds_t *)(p3_smem + Q3_RING + (region) * Q3_PAR + 2048 + (wave & 3) * 256), 4, 0, 0); \
    }                                                                                                              \
  }

  // ---- fragment addresses ----------------------------------------------------------------------------------------------
  const int pz = ((lane >> 4) ^ ((lane >> 1) & 7)) * 16;
  const int a_rd = (wm * 32 + (lane & 15)) * 128 + pz;         // A-lo / A-hi image: + i * 2048, mid plane ^ 64
  const int b_rd = 2 * P3_SLOT + (wn * 64 + (lane & 15)) * 128 + pz;   // B image: + j * 2048
  int cb = 0;                                                  // ring buffer (byte offset) being read
  const unsigned lds0 = (unsigned)(uintptr_t)(p3_lds_t *)p3_smem;

  // `res`: the previous tile's 16 accumulator tiles; p_ia / p_iw: that tile's row scales (4 row tiles) and column scales (4
  // column tiles x 4) for this lane, read from the parameter region ONCE per tile (per piece it was 245 cycles of LDS latency
  // in the load segment); a piece's bias is requested when the piece falls due and used after the phase's fragment reads and
  // DMA requests (holding it too would take 16 more registers than there are).  All indices into them are static: the k-tiles
  // that carry pieces are unrolled (a runtime index puts the array into scratch memory; a switch over 16 cases measured 380
  // cycles of scalar branches per piece; two 32-float vectors under a uniform dynamic index spilled 14-20 registers).
  hf32x4 acc[4][4], res[4][4];
  float p_ia[4] = {0.f, 0.f, 0.f, 0.f};
  hf32x4 p_iw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p_iw[i] = hf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j] = hf32x4{0.f, 0.f, 0.f, 0.f}; res[i][j] = hf32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  hf16x8 fa[2][2], fb[4][2];
  bool has_res = false;
  // the result as a buffer: a lane outside it (or a piece before the first tile is done) stores at an offset the hardware drops
  const __amdgpu_buffer_rsrc_t drs = PLANES_OUT
      ? __builtin_amdgcn_make_buffer_rsrc(a.Dp, 0, (int)((unsigned)a.M * (unsigned)(((a.N + 31) & ~31) * 4)), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(a.D, 0, (int)((unsigned)a.M * (unsigned)a.ldd * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(PLANES_OUT ? a.d_inv : nullptr, 0, PLANES_OUT ? a.M * 4 : 0, 0x00020000);
  const float d_bound0 = PLANES_OUT ? a.d_bound[0] : 0.f, d_bound1 = PLANES_OUT ? a.d_bound[1] : 0.f;
  int rm0 = 0, rn0 = 0, rpar = 0;                              // origin and parameter region of the tile `res` belongs to
  int cpar = 0;                                                // parameter region of the tile being accumulated

#define Q3_READ_A(slice)                                                                                           \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
    fa[i][0] = *reinterpret_cast<const hf16x8 *>(p3_smem + cb + (slice) * P3_SLOT + i * 2048 + a_rd);              \
    fa[i][1] = *reinterpret_cast<const hf16x8 *>(p3_smem + cb + (slice) * P3_SLOT + i * 2048 + (a_rd ^ 64));       \
  }
#define Q3_READ_B()                                                                                                \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
    fb[j][0] = *reinterpret_cast<const hf16x8 *>(p3_smem + cb + j * 2048 + b_rd);                                  \
    fb[j][1] = *reinterpret_cast<const hf16x8 *>(p3_smem + cb + j * 2048 + (b_rd ^ 64));                           \
  }
#define Q3_MMA(ah)                                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {                    \
    hf32x4 c = acc[(ah) * 2 + i][j];                                                                               \
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][1], fa[i][0], c, 0, 0, 0);                                    \
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][1], c, 0, 0, 0);                                    \
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][0], fa[i][0], c, 0, 0, 0);                                    \
    acc[(ah) * 2 + i][j] = c;                                                                                      \
  }
  // one activation of this lane's four values of accumulator tile (i, j) of `res`
#define Q3_VAL(v, i, j, b4)                                                                                        \
  {                                                                                                                \
    const hf32x4 c = res[i][j];                                                                                    \
    const float ia = p_ia[i];                                                                                      \
    const hf32x4 iw = p_iw[j];                                                                                     \
    v.x = (c[0] * ia) * iw[0] + b4[0]; v.y = (c[1] * ia) * iw[1] + b4[1];                                          \
    v.z = (c[2] * ia) * iw[2] + b4[2]; v.w = (c[3] * ia) * iw[3] + b4[3];                                          \
    if (ACT == 1) {                                                                                                \
      const isg_f32x2 g0 = gelu_exact2(isg_f32x2{v.x, v.y}), g1 = gelu_exact2(isg_f32x2{v.z, v.w});                \
      v.x = g0.x; v.y = g0.y; v.z = g1.x; v.w = g1.y;                                                              \
    }                                                                                                              \
    if (ACT == 2) {                                                                                                \
      v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; \
    }                                                                                                              \
  }
  // A piece = one accumulator tile of `res` (planes32 result: a pair of them).  Its ARITHMETIC (scales, bias, activation, the
  // split) is straight-line vector code that sits inside an MFMA segment -- the matrix core runs 16 cycles per MFMA and
  // takes 4 to issue, the vector instructions fill the gaps; in the load segment (where it sat first) its 390 cycles made
  // that segment twice as long as the MFMA segment of the other wave group, which then waited at the barrier
  // (profiles/r04_h_h3p_stamps.txt).  Its STORE is a buffer store issued after the segment's last MFMA: lanes outside D
  // get an offset beyond the descriptor's range and the hardware drops them -- no branch, and EVERY wave issues every
  // store, so the counted waits need no stand-in requests.
  // fp32 result: the finished values of accumulator tile (i, j); the byte offset of this lane's 16 bytes (0xFFFFFFF0: dropped)
#define Q3_EPI_VAL(v, i, j, b4) { float4 t_; Q3_VAL(t_, i, j, b4) v = hf32x4{t_.x, t_.y, t_.z, t_.w}; }
#define Q3_EPI_OFF(off, i, j, live)                                                                                \
  {                                                                                                                \
    const int rl = wm * 64 + (i) * 16 + (lane & 15), cl = wn * 64 + (j) * 16 + 4 * (lane >> 4);                    \
    const bool ok = (int)(live) & (int)(rm0 + rl < a.M) & (int)(rn0 + cl < a.N);   /* no branches */                    \
    off = ok ? ((unsigned)(rm0 + rl) * (unsigned)a.ldd + (unsigned)(rn0 + cl)) * 4u : 0xFFFFFFF0u;                 \
  }
  // `a.nt_store`: the cache policy of a large result's stores, uniform: 0 plain, 1 nt, 2 sc0 sc1 nt (write-through, streaming).
  // A 128-KB-per-tile result stream through a 4 MB L2 pushes the weights out of it.  Measured (profiles/r04_ag_h3p_store_policy.txt):
  // alone, policy 2 runs the K = 300 shapes 15-32 % faster than 0 on some MI355X boxes and 20 % slower on others; inside the full
  // model every policy gives the same step, so the default (isg_linear_h3p_store_policy(-1)) is 1 at K >= 512, 0 below.
#define Q3_EPI_ST(v, off)                                                                                          \
  {                                                                                                                \
    if (a.nt_store == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p3_u32x4, v), drs, (int)(off), 0, 19); \
    else if (a.nt_store == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p3_u32x4, v), drs, (int)(off), 0, 2); \
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p3_u32x4, v), drs, (int)(off), 0, 0);           \
  }
  // planes32 result: the PAIR of accumulator tiles (i, j0), (i, j0 + 1) = one 32-column group of 16 rows.  A lane holds
  // columns 4q .. 4q + 3 of each tile (q = lane >> 4): after the split it swaps ONE 8-byte fragment with lane ^ 16, so that it
  // owns 8 consecutive columns of the group -- 16 bytes of the hi plane and 16 of the mid plane, four lanes = the row's 64
  // contiguous bytes per plane: two stores per lane and pair, the store count and width of the fp32 path (the first form
  // stored two 8-byte pieces per tile: 380 us where the fp32 result took 246).  Columns in [N, roundup32(N)) are the NEXT
  // Linear's k padding: written as zeros.
#define Q3_PAIR_VAL(h16, m16, i, j0, b4a, b4b, ibnd)                                                      \
  {                                                                                                                \
    const int q = lane >> 4, cl = wn * 64 + (j0) * 16 + 4 * q;                                                     \
    float4 va, vb;                                                                                                 \
    Q3_VAL(va, i, j0, b4a)                                                                                         \
    Q3_VAL(vb, i, (j0) + 1, b4b)                                                                                   \
    if (rn0 + cl >= a.N) va = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
    if (rn0 + cl + 16 >= a.N) vb = make_float4(0.f, 0.f, 0.f, 0.f);                                                \
    float so, inv;                                                                                                 \
    h3_scale((ibnd) * d_bound0 + d_bound1, so, inv);                                                               \
    va.x *= so; va.y *= so; va.z *= so; va.w *= so; vb.x *= so; vb.y *= so; vb.z *= so; vb.w *= so;                \
    const hf16x4 ha = {(_Float16)va.x, (_Float16)va.y, (_Float16)va.z, (_Float16)va.w};                            \
    const hf16x4 hb = {(_Float16)vb.x, (_Float16)vb.y, (_Float16)vb.z, (_Float16)vb.w};                            \
    const hf16x4 ma = {(_Float16)(va.x - (float)ha[0]), (_Float16)(va.y - (float)ha[1]),                           \
                       (_Float16)(va.z - (float)ha[2]), (_Float16)(va.w - (float)ha[3])};                          \
    const hf16x4 mb = {(_Float16)(vb.x - (float)hb[0]), (_Float16)(vb.y - (float)hb[1]),                           \
                       (_Float16)(vb.z - (float)hb[2]), (_Float16)(vb.w - (float)hb[3])};                          \
    typedef int p3_i32x2 __attribute__((ext_vector_type(2)));                                                      \
    const bool even = (q & 1) == 0;       /* even q keeps its tile-a fragment and sends tile b's; odd q the reverse */ \
    const p3_i32x2 hka = __builtin_bit_cast(p3_i32x2, ha), hkb = __builtin_bit_cast(p3_i32x2, hb);                 \
    const p3_i32x2 mka = __builtin_bit_cast(p3_i32x2, ma), mkb = __builtin_bit_cast(p3_i32x2, mb);                 \
    const p3_i32x2 hs = even ? hkb : hka, ms = even ? mkb : mka;                                                   \
    const p3_i32x2 hr = {__shfl_xor(hs[0], 16, 64), __shfl_xor(hs[1], 16, 64)};                                    \
    const p3_i32x2 mr = {__shfl_xor(ms[0], 16, 64), __shfl_xor(ms[1], 16, 64)};                                    \
    h16 = even ? p3_i32x4{hka[0], hka[1], hr[0], hr[1]} : p3_i32x4{hr[0], hr[1], hkb[0], hkb[1]};                  \
    m16 = even ? p3_i32x4{mka[0], mka[1], mr[0], mr[1]} : p3_i32x4{mr[0], mr[1], mkb[0], mkb[1]};                  \
  }
#define Q3_PAIR_OFF(off, ioff, inv, i, j0, live, ibnd)                                                                   \
  {                                                                                                                \
    const int npad = (a.N + 31) & ~31;                                                                             \
    const int q = lane >> 4, rl = wm * 64 + (i) * 16 + (lane & 15);                                                \
    /* chunk of 8 columns this lane owns: q = 0, 1, 2, 3 -> columns 0-7, 16-23, 8-15, 24-31 of the group */          \
    const int chunk = ((q & 1) << 1) | (q >> 1);                                                                   \
    const int gcol = rn0 + wn * 64 + (j0) * 16;                      /* first column of the 32-column group */     \
    const bool ok = (int)(live) & (int)(rm0 + rl < a.M) & (int)(gcol + chunk * 8 < npad);                          \
    off = ok ? ((unsigned)(rm0 + rl) * (unsigned)(npad * 2) + (unsigned)((gcol >> 5) * 64 + chunk * 8)) * 2u : 0xFFFFFFF0u; \
    ioff = ((int)ok & (int)(gcol == 0) & (int)(q == 0)) ? (unsigned)(rm0 + rl) * 4u : 0xFFFFFFF0u;                 \
    float so_;                                                                                                     \
    h3_scale((ibnd) * d_bound0 + d_bound1, so_, inv);                                                              \
  }
  // (the row's inverse scale leaves with the row's first column group -- ioff is valid in that pair's lanes only: one more
  // store in the waves of the first column tile; the counted waits then allow one operation fewer in flight than there are,
  // i.e. they wait a little early, never late)
#define Q3_PAIR_ST(h16, m16, off, ioff, inv)                                                                       \
  {                                                                                                                \
    const unsigned off2_ = (off) == 0xFFFFFFF0u ? 0xFFFFFFF0u : (off) + 64u;                                       \
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p3_u32x4, h16), drs, (int)(off), 0, 0);              \
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p3_u32x4, m16), drs, (int)off2_, 0, 0);              \
    if (rn0 == 0 && wn == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(inv), irs, (int)(ioff), 0, 0); \
  }
  // the tile's accumulators become `res`; its scales come out of the parameter region the DMA filled a tile ago.  Read by
  // inline asm: hipcc (ROCm 7.2) orders a VISIBLE LDS load behind every outstanding 4-byte LDS-DMA it cannot tell apart
  // from the load's address -- `s_waitcnt vmcnt(0)`, the whole ring drained.
#define Q3_TILE_END()                                                                                              \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {                  \
      res[i][j] = acc[i][j];                                                                                       \
      acc[i][j] = hf32x4{0.f, 0.f, 0.f, 0.f};                                                                      \
    }                                                                                                              \
    const unsigned pa = lds0 + Q3_RING + cpar * Q3_PAR;                                                            \
    const unsigned par = pa + (wm * 64 + (lane & 15)) * 4, pac = pa + 1024 + (wn * 64 + 4 * (lane >> 4)) * 4;      \
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:64\n\tds_read_b32 %2, %8 offset:128\n\t"         \
                 "ds_read_b32 %3, %8 offset:192\n\t"                                                               \
                 "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:64\n\tds_read_b128 %6, %9 offset:128\n\t"      \
                 "ds_read_b128 %7, %9 offset:192\n\t"                                                              \
                 "s_waitcnt lgkmcnt(0)"                                                                            \
                 : "=&v"(p_ia[0]), "=&v"(p_ia[1]), "=&v"(p_ia[2]), "=&v"(p_ia[3]), "=&v"(p_iw[0]), "=&v"(p_iw[1]), \
                   "=&v"(p_iw[2]), "=&v"(p_iw[3])                                                                  \
                 : "v"(par), "v"(pac) : "memory");                                                                 \
  }
  // a piece's bias: requested from the parameter region of the tile `res` belongs to (no wait here)
#define Q3_BIAS_REQ(bq, j)                                                                                         \
  asm volatile("ds_read_b128 %0, %1 offset:1536" : "=&v"(bq)                                                       \
               : "v"(lds0 + Q3_RING + rpar * Q3_PAR + (wn * 64 + (j) * 16 + 4 * (lane >> 4)) * 4) : "memory");
  // segmented A, planes32 result: the row's LARGER inverse scale (what the result's bound needs), left in the a_inv0 slot of the
  // tile's parameter region by Q3_RESCALE (its own slot: the two wave groups pass the boundary a segment apart and each
  // needs a_inv0 intact)
#define Q3_BND_REQ(ib, i)                                                                                          \
  asm volatile("ds_read_b32 %0, %1 offset:3072" : "=&v"(ib)                                                        \
               : "v"(lds0 + Q3_RING + rpar * Q3_PAR + (wm * 64 + (i) * 16 + (lane & 15)) * 4) : "memory");
  // k-tile kseg of a segmented A operand: the accumulators pass from units of a_inv0 to units of a_inv (see SEG above)
#define Q3_RESCALE()                                                                                               \
  {                                                                                                                \
    const unsigned pa_ = lds0 + Q3_RING + cpar * Q3_PAR + (wm * 64 + (lane & 15)) * 4;                             \
    float i1_[4], i0_[4];                                                                                          \
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:64\n\tds_read_b32 %2, %8 offset:128\n\t"          \
                 "ds_read_b32 %3, %8 offset:192\n\t"                                                               \
                 "ds_read_b32 %4, %8 offset:2048\n\tds_read_b32 %5, %8 offset:2112\n\tds_read_b32 %6, %8 offset:2176\n\t" \
                 "ds_read_b32 %7, %8 offset:2240\n\t"                                                              \
                 "s_waitcnt lgkmcnt(0)"                                                                            \
                 : "=&v"(i1_[0]), "=&v"(i1_[1]), "=&v"(i1_[2]), "=&v"(i1_[3]), "=&v"(i0_[0]), "=&v"(i0_[1]),       \
                   "=&v"(i0_[2]), "=&v"(i0_[3])                                                                    \
                 : "v"(pa_) : "memory");                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
      /* both are powers of two: their quotient by exponent arithmetic, exact */                                   \
      const float r_ = __uint_as_float(__float_as_uint(i0_[i]) - __float_as_uint(i1_[i]) + 0x3F800000u);           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] *= r_;                                               \
      i0_[i] = fmaxf(i0_[i], i1_[i]);                                                                              \
    }                                                                                                              \
    asm volatile("ds_write_b32 %4, %0 offset:3072\n\tds_write_b32 %4, %1 offset:3136\n\tds_write_b32 %4, %2 offset:3200\n\t" \
                 "ds_write_b32 %4, %3 offset:3264\n\t"                                                             \
                 "s_waitcnt lgkmcnt(0)"                                                                            \
                 :: "v"(i0_[0]), "v"(i0_[1]), "v"(i0_[2]), "v"(i0_[3]), "v"(pa_) : "memory");                      \
  }
#define Q3_BAR                                                                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  ISG_BARRIER();                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define Q3_WAIT(n)                                                                                                 \
  ISG_WAIT(0x0F70 | ((n) & 15) | (((n) >> 4) << 14));
#define Q3_M(ah)                                                                                                   \
  ISG_WAIT(0xC07F);                                                                              \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  Q3_ST(3 + 7 * (ah))                                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                                   \
  Q3_MMA(ah)                                                                                                       \
  __builtin_amdgcn_s_setprio(0);                                                                                   \
  Q3_ST(4 + 7 * (ah))                                                                                              \
  Q3_BAR                                                                                                           \
  Q3_ST(5 + 7 * (ah))
  // The same with a piece of `res` inside: its vector arithmetic is part of the MFMA segment's scheduling region (the matrix
  // core takes a new MFMA every 16 cycles and an issue costs 4: the vector instructions fit the gaps), its store(s) follow the
  // last MFMA.  TI / TJ: the accumulator tile (fp32) or the pair's first tile (planes32); BQ0 / BQ1: its bias, requested in
  // the load segment before.  (Also built and measured, same-box A/B in DESIGN 15.1: the finished piece held in its registers
  // and stored behind the NEXT load segment's requests -- every segment then issues a store, dropped when nothing is pending,
  // and the K = 2048 shape lost 7 % to those.)
#define Q3_M_PIECE(ah, TI, TJ, BQ0, BQ1, IB)                                                                        \
  ISG_WAIT(0xC07F);                                                                              \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  Q3_ST(3 + 7 * (ah))                                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                                   \
  if constexpr (PLANES_OUT) {                                                                                      \
    p3_i32x4 h16_, m16_;                                                                                           \
    unsigned off_, ioff_;                                                                                          \
    float inv_;                                                                                                    \
    Q3_PAIR_VAL(h16_, m16_, TI, TJ, BQ0, BQ1, IB)                                                                  \
    Q3_PAIR_OFF(off_, ioff_, inv_, TI, TJ, has_res, IB)                                                            \
    Q3_MMA(ah)                                                                                                     \
    Q3_INTERLEAVE()                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    Q3_PAIR_ST(h16_, m16_, off_, ioff_, inv_)                                                                      \
  } else {                                                                                                         \
    hf32x4 v_;                                                                                                     \
    unsigned off_;                                                                                                 \
    Q3_EPI_VAL(v_, TI, TJ, BQ0)                                                                                    \
    Q3_EPI_OFF(off_, TI, TJ, has_res)                                                                              \
    Q3_MMA(ah)                                                                                                     \
    Q3_INTERLEAVE()                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    Q3_EPI_ST(v_, off_)                                                                                            \
  }                                                                                                                \
  Q3_ST(4 + 7 * (ah))                                                                                              \
  Q3_BAR                                                                                                           \
  Q3_ST(5 + 7 * (ah))
  // one MFMA, then up to three of the piece's vector instructions, 24 times (fp32 result).  The planes32 pair's arithmetic
  // (two tiles, the split, the lane exchange) interleaved this way needs more registers than there are (1-2 spilled -- and a
  // spill is a scratch access, i.e. a vector-memory operation the counted waits know nothing of): it stays in one block
#ifndef ISG_Q3_VALU_PER_MFMA
#define ISG_Q3_VALU_PER_MFMA 3
#endif
#define Q3_INTERLEAVE()                                                                                            \
  _Pragma("unroll") for (int z = 0; z < 24; ++z) {                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
    __builtin_amdgcn_sched_group_barrier(0x002, PLANES_OUT ? 0 : ISG_Q3_VALU_PER_MFMA, 0);                         \
  }
  // End of a load segment: every request is waited for FOUR segments after its issue (B and A-lo leave in the first load
  // segment of a k-tile and are read two k-tiles later in a first segment, A-hi likewise in second segments).  A wave's
  // operations in issue order, k-tile c:  P0(c): 4 requests | S0(c): stores of the piece in MFMA segment 0 | P1(c): pc(c)
  // parameter request + 2 requests | S1(c): stores of the piece in MFMA segment 1.  The wait at the end of P0(c) needs P1(c - 2)
  // landed, so everything younger may be in flight: S1(c-2) + 4 + S0(c-1) + 2 + pc(c-1) + S1(c-1) + 4; the wait at the end of
  // P1(c) needs P0(c - 1): S0(c-1) + 2 + pc(c-1) + S1(c-1) + 4 + S0(c) + 2 + pc(c).  A term whose k-tile lies before the head
  // (the tile before) is taken as 0, which only waits earlier than needed.
  //   fp32 result: HEAD 16 -> one tile per k-tile, in segment 1; HEAD 8 -> tiles 2 kh (segment 0) and 2 kh + 1 (segment 1)
  //   planes32:    one PAIR (two stores) in segment 1 -- of every k-tile at HEAD 8, of the odd ones at HEAD 16
  // (a workgroup's first tile has no `res` yet: its pieces run on zeros and their stores are dropped like any lane outside D)
#define Q3_S0(kh, NP, hd) (((hd) && (kh) >= 0 && !PLANES_OUT && (NP) == 2) ? 1 : 0)
#define Q3_S1(kh, NP, hd) (((hd) && (kh) >= 0) ? (PLANES_OUT ? (((NP) == 2 || ((kh) & 1)) ? 2 : 0) : 1) : 0)
#define Q3_PC(kh, hd) (((hd) && (kh) == 0) ? (SEG ? 2 : 1) : 0)
#define Q3_KT(kh, NP, is_head)                                                                                     \
  {                                                                                                                \
    constexpr int w0s = Q3_S1((kh) - 2, NP, is_head) + Q3_S0((kh) - 1, NP, is_head) + Q3_S1((kh) - 1, NP, is_head); \
    constexpr int w0p = Q3_PC((kh) - 1, is_head);                                                                  \
    constexpr int w1s = Q3_S0((kh) - 1, NP, is_head) + Q3_S1((kh) - 1, NP, is_head) + Q3_S0(kh, NP, is_head);      \
    constexpr int w1p = Q3_PC((kh) - 1, is_head) + Q3_PC(kh, is_head);                                             \
    constexpr bool pc0 = Q3_S0(kh, NP, is_head) > 0, pc1 = Q3_S1(kh, NP, is_head) > 0;                             \
    /* the accumulator tile(s) that leave in this k-tile */                                                        \
    constexpr int ta = PLANES_OUT ? 0 : (NP) * (kh);                             /* segment 0 (fp32, HEAD 8) */     \
    constexpr int tb = PLANES_OUT ? ((NP) == 2 ? 2 * (kh) : ((kh) & ~1)) : ((NP) * (kh) + (NP) - 1);   /* segment 1 */ \
    hf32x4 bq0 = {0.f, 0.f, 0.f, 0.f}, bq1 = {0.f, 0.f, 0.f, 0.f};                                                 \
    Q3_READ_A(0) Q3_READ_B()                                                                                       \
    if (pc0 && a.bias) Q3_BIAS_REQ(bq0, ta & 3)                                                                    \
    Q3_STAGE_P0()                                                                                                  \
    Q3_ST(0)                                                                                                       \
    Q3_WAIT(10 + w0s + w0p)                                                                                        \
    Q3_ST(1)                                                                                                       \
    Q3_BAR                                                                                                         \
    Q3_ST(2)                                                                                                       \
    if constexpr (pc0) { Q3_M_PIECE(0, (ta & 15) >> 2, ta & 3, bq0, bq0, p_ia[(ta & 15) >> 2]) } else { Q3_M(0) }  \
    Q3_READ_A(1)                                                                                                   \
    if (pc1 && a.bias) {                                                                                           \
      Q3_BIAS_REQ(bq0, tb & 3)                                                                                     \
      if (PLANES_OUT) Q3_BIAS_REQ(bq1, (tb + 1) & 3)                                                               \
    }                                                                                                              \
    float ib_ = p_ia[(tb & 15) >> 2];                                                                              \
    if constexpr (SEG && PLANES_OUT && pc1) Q3_BND_REQ(ib_, (tb & 15) >> 2)                                        \
    if ((is_head) && (kh) == 0) Q3_PARAMS(nm0, nn0, npar)                                                          \
    Q3_STAGE_P1()                                                                                                  \
    Q3_ST(7)                                                                                                       \
    Q3_ST(6)                                                                                                       \
    Q3_WAIT(8 + w1s + w1p)                                                                                         \
    Q3_ST(8)                                                                                                       \
    Q3_BAR                                                                                                         \
    Q3_ST(9)                                                                                                       \
    if constexpr (pc1) { Q3_M_PIECE(1, (tb & 15) >> 2, tb & 3, bq0, bq1, ib_) } else { Q3_M(1) }                   \
    cb = cb == 2 * Q3_BUF ? 0 : cb + Q3_BUF;                                                                       \
  }

  // ---- prologue: the first tile's parameters, stream k-tiles 0 and 1 -----------------------------------------------------
  Q3_PARAMS(cm0, cn0, 0)
  Q3_STAGE_P0()
  Q3_STAGE_P1()
  Q3_STAGE_P0()
  Q3_STAGE_P1()
  Q3_WAIT(6)
  Q3_BAR
  if (wn == 1) { Q3_BAR }                // the second wave group runs one barrier behind the first

  ISG_DIAG_BEGIN()
  while (true) {
    // the tile after this one (for its parameters; the stager finds it by itself)
    int nm0 = cm0, nn0 = cn0;
    const int nL = next_tile(cL, nm0, nn0);
    if (nL < 0) { nm0 = cm0; nn0 = cn0; }
    const int npar = cpar == 2 ? 0 : cpar + 1;
    if constexpr (HEAD == 16) {
      Q3_KT(0, 1, true) Q3_KT(1, 1, true) Q3_KT(2, 1, true) Q3_KT(3, 1, true) Q3_KT(4, 1, true) Q3_KT(5, 1, true)
      Q3_KT(6, 1, true) Q3_KT(7, 1, true) Q3_KT(8, 1, true) Q3_KT(9, 1, true) Q3_KT(10, 1, true) Q3_KT(11, 1, true)
      Q3_KT(12, 1, true) Q3_KT(13, 1, true) Q3_KT(14, 1, true) Q3_KT(15, 1, true)
    } else {
      Q3_KT(0, 2, true) Q3_KT(1, 2, true) Q3_KT(2, 2, true) Q3_KT(3, 2, true) Q3_KT(4, 2, true) Q3_KT(5, 2, true)
      Q3_KT(6, 2, true) Q3_KT(7, 2, true)
    }
#pragma unroll 1
    for (int kt = HEAD; kt < nk; ++kt) {
      if constexpr (SEG) if (kt == a.kseg) Q3_RESCALE()
      Q3_KT(0, 0, false)
    }
    Q3_TILE_END()
    has_res = true;
    rm0 = cm0; rn0 = cn0; rpar = cpar;
    if (nL < 0) break;
    cL = nL; cm0 = nm0; cn0 = nn0; cpar = npar;
  }
  if (wn == 0) { Q3_BAR }
  ISG_WAIT(0x0F70);    // the requests past the last k-tile have landed: nothing in flight at the end
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      hf32x4 bqa = {0.f, 0.f, 0.f, 0.f}, bqb = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
        Q3_BIAS_REQ(bqa, j)
        Q3_BIAS_REQ(bqb, j + 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (PLANES_OUT) {
        p3_i32x4 h16_, m16_;
        unsigned off_, ioff_;
        float inv_, ib_ = p_ia[i];
        if constexpr (SEG) {
          Q3_BND_REQ(ib_, i)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        Q3_PAIR_VAL(h16_, m16_, i, j, bqa, bqb, ib_)
        Q3_PAIR_OFF(off_, ioff_, inv_, i, j, true, ib_)
        Q3_PAIR_ST(h16_, m16_, off_, ioff_, inv_)
      } else {
        hf32x4 va_, vb_;
        unsigned oa_, ob_;
        Q3_EPI_VAL(va_, i, j, bqa)
        Q3_EPI_VAL(vb_, i, j + 1, bqb)
        Q3_EPI_OFF(oa_, i, j, true)
        Q3_EPI_OFF(ob_, i, j + 1, true)
        Q3_EPI_ST(va_, oa_)
        Q3_EPI_ST(vb_, ob_)
      }
    }
  ISG_DIAG_DUMP(g_p3_stamps, blockIdx.x * 8 + wave, 15, )
#undef Q3_KT
#undef Q3_TILE_END
#undef Q3_BIAS_REQ
#undef Q3_BND_REQ
#undef Q3_RESCALE
#undef Q3_STORES
#undef Q3_PAIR_ST
#undef Q3_PAIR_VAL
#undef Q3_PAIR_OFF
#undef Q3_EPI_OFF
#undef Q3_M_PIECE
#undef Q3_INTERLEAVE
#undef Q3_S0
#undef Q3_S1
#undef Q3_PC
#undef Q3_VAL
#undef Q3_M
#undef Q3_WAIT
#undef Q3_BAR
#undef Q3_EPI_ST
#undef Q3_EPI_VAL
#undef Q3_MMA
#undef Q3_READ_B
#undef Q3_READ_A
#undef Q3_PARAMS
#undef Q3_STAGE_P1
#undef Q3_STAGE_P0
#undef Q3_DMA
#undef Q3_SOFF
}

}  // namespace isg

using namespace isg;

#include <atomic>
static std::atomic<int> g_h3p_store_policy{-1};

// The cache policy of a LARGE (>= 128 MB) fp32 result's stores for the rest of the process: 0 plain, 1 nt, 2 sc0 sc1 nt, -1 the
// built-in choice (nt at K >= 512, plain below); ISG_EINVAL on anything else.  An experiment's switch (see Q3_EPI_ST).
extern "C" int isg_linear_h3p_store_policy(int32_t policy) {
  if (policy < -1 || policy > 2) return ISG_EINVAL;
  g_h3p_store_policy.store(policy, std::memory_order_relaxed);
  return ISG_OK;
}

ISG_DIAG_SETTER(isg_p3_set_stamp_buffer, g_p3_stamps)

extern "C" int64_t isg_planes32_elems(int64_t rows, int32_t K) {
  if (rows <= 0 || K <= 0) return 0;
  return rows * (int64_t)((K + 31) / 32) * 64;
}

// planes: uint16[isg_planes32_elems(M, K)]; inv_scale: float[M].  K % 4 == 0, lda % 4 == 0, a 16-byte aligned.
extern "C" int isg_split_planes32(const float *a, int64_t M, int32_t K, int32_t lda, uint16_t *planes, float *inv_scale,
                                  void *stream) {
  if (M < 0 || K <= 0 || lda < K) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a || !planes || !inv_scale) return ISG_EINVAL;
  if ((K & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15) ||
      (M + 3) / 4 >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  ISG_PLANES32_ROWS_LAUNCH(split_planes32_kernel, M, K >> 2, as_stream(stream), a, (int)M, K, lda,
                           reinterpret_cast<_Float16 *>(planes), inv_scale);
  return check_launch();
}

// rows fp32 [N, C] or NULL; planes: uint16[isg_planes32_elems(N, C)]; inv_scale: float[N].  C % 4 == 0, 16-byte aligned rows.
extern "C" int isg_instr_gate_planes32(const float *x, const float *instr, const int64_t *batch, float *rows, uint16_t *planes,
                                       float *inv_scale, int64_t N, int32_t C, void *stream) {
  if (N < 0 || C <= 0) return ISG_EINVAL;
  if (N == 0) return ISG_OK;
  if (!x || !instr || !batch || !planes || !inv_scale) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if ((C & 3) || mis(x) || mis(instr) || mis(planes) || (rows && mis(rows)) || (N + 3) / 4 >= (1ll << 31)) return ISG_EUNSUPPORTED;
  ISG_PLANES32_ROWS_LAUNCH(instr_gate_planes32_kernel, N, C >> 2, as_stream(stream), x, instr,
                           reinterpret_cast<const long long *>(batch), (int)N, C, rows, reinterpret_cast<_Float16 *>(planes), inv_scale);
  return check_launch();
}

// Exactly one output form: d (fp32 [M, ldd]) or d_planes + d_inv + d_bound (planes32 of the result, N % 32 == 0).
// ISG_EUNSUPPORTED: N % 4, ldd % 4, misaligned pointers, an operand of 2 GB or more.
extern "C" int isg_linear_h3p(const uint16_t *a_planes, const float *a_inv, const uint16_t *w_planes, const float *w_inv,
                              const float *bias, float *d, uint16_t *d_planes, float *d_inv, const float *d_bound, int64_t M,
                              int32_t N, int32_t K, int32_t ldd, int32_t act, const float *a_inv_first, int32_t k_split,
                              void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || act < 0 || act > 2 || k_split < 0 || (k_split & 31) || k_split >= K ||
      (a_inv_first != nullptr) != (k_split > 0))
    return ISG_EINVAL;
  const bool planes_out = d == nullptr;
  if (planes_out ? (!d_planes || !d_inv || !d_bound) : (d_planes || d_inv || ldd < N)) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a_planes || !a_inv || !w_planes || !w_inv) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  const int KT = (K + 31) / 32;
  if ((N & 3) || mis(a_planes) || mis(w_planes) || mis(w_inv) || (bias && mis(bias)) ||
      (planes_out ? mis(d_planes) : ((ldd & 3) || mis(d))) ||
      M * (int64_t)KT * 128 >= (1ll << 31) || (int64_t)N * KT * 128 >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  const long long tm = (M + P3_T - 1) / P3_T, tn = (N + P3_T - 1) / P3_T;
  const long long blocks = tn * ((tm + 7) / 8 * 8);
  if (blocks >= (1ll << 31)) return ISG_EUNSUPPORTED;
  static const long long nt_mb = [] { const char *e = getenv("ISG_GEMM_NT_MB"); return e ? atoll(e) : 128ll; }();
  const bool large = nt_mb >= 0 && (long long)M * N * 4 >= nt_mb * 1000000ll;
  const int forced = g_h3p_store_policy.load(std::memory_order_relaxed);
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  P3Args a = {
      .A = reinterpret_cast<const _Float16 *>(a_planes), .W = reinterpret_cast<const _Float16 *>(w_planes), .a_inv = a_inv,
      .w_inv = w_inv, .bias = bias, .D = d, .Dp = reinterpret_cast<_Float16 *>(d_planes), .d_inv = d_inv, .d_bound = d_bound,
      .M = (int)M, .N = N, .KT = KT, .ldd = ldd, .tiles_n = (int)tn,
      .nt_store = !large || planes_out ? 0 : forced >= 0 ? forced : (KT >= 16 ? 1 : 0),      // a planes32 result: plain (measured)
      .a_inv0 = a_inv_first, .kseg = k_split >> 5};
  if (!a.A || !a.W || !a.a_inv || !a.w_inv || (a.kseg && !a.a_inv0) || (planes_out ? (!a.Dp || !a.d_inv || !a.d_bound) : !a.D))
    return ISG_EINVAL;                         // the struct the kernels dereference, not the parameters it was filled from
  hipStream_t st = as_stream(stream);
  static const int version = [] { const char *e = getenv("ISG_H3P_V"); return e ? atoi(e) : 2; }();
  if (planes_out && (N & 31) && (version == 1 || KT < Q3_HEAD)) return ISG_EUNSUPPORTED;   // the 256 x 256 form pads no columns
  // a segmented A operand: the persistent form's instantiations with an exact GELU (x_proj.0 of mgat.py:156), the boundary behind
  // the k-tiles that carry result pieces
  if (a.kseg && (version == 1 || KT < 16 || a.kseg < 16 || act != 1)) return ISG_EUNSUPPORTED;
  if (version != 1 && KT >= Q3_HEAD) {       // persistent 256 x 128 form
    // the result is addressed through a buffer descriptor with 32-bit byte offsets
    if ((planes_out ? M * (int64_t)(((N + 31) & ~31) * 4) : M * (int64_t)ldd * 4) >= (1ll << 32) - 16) return ISG_EUNSUPPORTED;
    const int ncu = device_cus();
    const long long tn2 = (N + Q3_BN - 1) / Q3_BN;
    const long long total = tn2 * ((tm + 7) / 8 * 8);
    if (total >= (1ll << 31)) return ISG_EUNSUPPORTED;
    Q3Args q = {.p = a, .tiles_m = (int)tm, .total_l = (int)total};
    q.p.tiles_n = (int)tn2;
    const unsigned grid = (unsigned)(total < ncu ? total : ncu);
#define ISG_Q3H(ACT_, PO_, H_)                                                                                    \
  do {                                                                                                            \
    if (!dyn_lds_ok<&linear_h3q_kernel<ACT_, PO_, H_>>(Q3_SMEM)) return ISG_ELAUNCH;                              \
    linear_h3q_kernel<ACT_, PO_, H_><<<grid, P3_THREADS, Q3_SMEM, st>>>(q);                                       \
  } while (0)
#define ISG_Q3(ACT_, PO_, slot) do { if (KT >= 16) ISG_Q3H(ACT_, PO_, 16); else ISG_Q3H(ACT_, PO_, 8); } while (0)
    if (a.kseg) {
      if (planes_out) {
        if (!dyn_lds_ok<&linear_h3q_kernel<1, true, 16, true>>(Q3_SMEM)) return ISG_ELAUNCH;
        linear_h3q_kernel<1, true, 16, true><<<grid, P3_THREADS, Q3_SMEM, st>>>(q);
      } else {
        if (!dyn_lds_ok<&linear_h3q_kernel<1, false, 16, true>>(Q3_SMEM)) return ISG_ELAUNCH;
        linear_h3q_kernel<1, false, 16, true><<<grid, P3_THREADS, Q3_SMEM, st>>>(q);
      }
    } else if (planes_out) {
      if (act == 1) ISG_Q3(1, true, 6); else if (act == 2) ISG_Q3(2, true, 7); else ISG_Q3(0, true, 8);
    } else {
      if (act == 1) ISG_Q3(1, false, 9); else if (act == 2) ISG_Q3(2, false, 10); else ISG_Q3(0, false, 11);
    }
#undef ISG_Q3
#undef ISG_Q3H
    return check_launch();
  }
#define ISG_P3(ACT_, PO_, slot)                                                                                   \
  do {                                                                                                            \
    if (!dyn_lds_ok<&linear_h3p_kernel<ACT_, PO_>>(P3_SMEM)) return ISG_ELAUNCH;                                  \
    linear_h3p_kernel<ACT_, PO_><<<(unsigned)blocks, P3_THREADS, P3_SMEM, st>>>(a);                               \
  } while (0)
  if (planes_out) {
    if (act == 1) ISG_P3(1, true, 0); else if (act == 2) ISG_P3(2, true, 1); else ISG_P3(0, true, 2);
  } else {
    if (act == 1) ISG_P3(1, false, 3); else if (act == 2) ISG_P3(2, false, 4); else ISG_P3(0, false, 5);
  }
#undef ISG_P3
  return check_launch();
}
