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

#include <stdlib.h>

namespace isg {

typedef __attribute__((address_space(3))) void p3_lds_t;
typedef __attribute__((address_space(1))) void p3_glb_t;

constexpr int P3_T = 256, P3_THREADS = 512;
constexpr int P3_SLOT = 128 * 128;          // one quarter image: 128 rows x (hi 32 | mid 32) fp16
constexpr int P3_BUF = 4 * P3_SLOT;         // A-lo, A-hi, B-lo, B-hi of one k-tile
constexpr int P3_SMEM = 2 * P3_BUF;         // 128 KB
constexpr int P3_ALO = 0, P3_AHI = 1, P3_BLO = 2, P3_BHI = 3;

// planes32 of fp32 rows: wave per row; the row's largest magnitude decides its scale
__global__ __launch_bounds__(256) void split_planes32_kernel(const float *__restrict__ a, int M, int K, int lda,
                                                             _Float16 *__restrict__ planes, float *__restrict__ inv_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int KT = (K + 31) >> 5, nc = K >> 2;
  const float4 *r4 = reinterpret_cast<const float4 *>(a + (int64_t)row * lda);
  float mx = 0.f;
  for (int c = lane; c < nc; c += 64) {
    const float4 v = r4[c];
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  mx = wave_max(mx);
  float s, inv;
  h3_scale(mx, s, inv);
  if (lane == 0) inv_out[row] = inv;
  _Float16 *p = planes + (int64_t)row * KT * 64;
  for (int c = lane; c < KT * 8; c += 64) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nc) v = r4[c];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    const hf16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    const hf16x4 mid = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                        (_Float16)(v.w - (float)hi[3])};
    _Float16 *d = p + (c >> 3) * 64 + (c & 7) * 4;
    *reinterpret_cast<hf16x4 *>(d) = hi;
    *reinterpret_cast<hf16x4 *>(d + 32) = mid;
  }
}

struct P3Args {
  const _Float16 *A, *W;          // planes32 [M][KT][64], [N][KT][64]
  const float *a_inv, *w_inv;     // [M], [N]
  const float *bias;              // [N] or NULL
  float *D;                       // fp32 [M, ldd], or NULL with the planes output below
  _Float16 *Dp;                   // planes32 [M][N / 32][64]
  float *d_inv;                   // [M]
  const float *d_bound;           // {2^14 * max_n ||W_n||_1, max |b|}: |D[m, :]| < inv_a[m] * d_bound[0] + d_bound[1]
  int M, N, KT, ldd, tiles_n, nt_store;
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
  const unsigned row_b = (unsigned)nk * 128u;          // bytes per operand row

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
  __builtin_amdgcn_s_waitcnt(0x0F76);   /* vmcnt(6) */                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_barrier();                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define P3_M(FB, ah, bh)                                                                                           \
  __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0) */                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_setprio(1);                                                                                   \
  P3_MMA(FB, ah, bh)                                                                                               \
  __builtin_amdgcn_s_setprio(0);                                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  __builtin_amdgcn_s_barrier();                                                                                    \
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
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int t = 0; t < nk; t += 2) {
    P3_TILE(t, 0)
    if (t + 1 < nk) P3_TILE(t + 1, 1)
  }
  if (wr == 0) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);    // the requests past the last k-tile have landed: nothing in flight at the end
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

}  // namespace isg

using namespace isg;

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
  split_planes32_kernel<<<(unsigned)((M + 3) / 4), 256, 0, as_stream(stream)>>>(a, (int)M, K, lda,
                                                                               reinterpret_cast<_Float16 *>(planes), inv_scale);
  return check_launch();
}

// Exactly one output form: d (fp32 [M, ldd]) or d_planes + d_inv + d_bound (planes32 of the result, N % 32 == 0).
// ISG_EUNSUPPORTED: N % 4, ldd % 4, misaligned pointers, an operand of 2 GB or more.
extern "C" int isg_linear_h3p(const uint16_t *a_planes, const float *a_inv, const uint16_t *w_planes, const float *w_inv,
                              const float *bias, float *d, uint16_t *d_planes, float *d_inv, const float *d_bound, int64_t M,
                              int32_t N, int32_t K, int32_t ldd, int32_t act, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return ISG_EINVAL;
  const bool planes_out = d == nullptr;
  if (planes_out ? (!d_planes || !d_inv || !d_bound) : (d_planes || d_inv || ldd < N)) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a_planes || !a_inv || !w_planes || !w_inv) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  const int KT = (K + 31) / 32;
  if ((N & 3) || mis(a_planes) || mis(w_planes) || mis(w_inv) || (bias && mis(bias)) ||
      (planes_out ? ((N & 31) || mis(d_planes)) : ((ldd & 3) || mis(d))) ||
      M * (int64_t)KT * 128 >= (1ll << 31) || (int64_t)N * KT * 128 >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  const long long tm = (M + P3_T - 1) / P3_T, tn = (N + P3_T - 1) / P3_T;
  const long long blocks = tn * ((tm + 7) / 8 * 8);
  if (blocks >= (1ll << 31)) return ISG_EUNSUPPORTED;
  P3Args a;
  a.A = reinterpret_cast<const _Float16 *>(a_planes); a.W = reinterpret_cast<const _Float16 *>(w_planes);
  a.a_inv = a_inv; a.w_inv = w_inv; a.bias = bias; a.D = d; a.Dp = reinterpret_cast<_Float16 *>(d_planes); a.d_inv = d_inv;
  a.d_bound = d_bound; a.M = (int)M; a.N = N; a.KT = KT; a.ldd = ldd; a.tiles_n = (int)tn;
  static const long long nt_mb = [] { const char *e = getenv("ISG_GEMM_NT_MB"); return e ? atoll(e) : 128ll; }();
  a.nt_store = nt_mb >= 0 && (long long)M * N * 4 >= nt_mb * 1000000ll;
  hipStream_t st = as_stream(stream);
  // more than 64 KB of dynamic LDS is an attribute of (function, device): set once per device of this process
  static bool attr_set[6][64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ISG_ELAUNCH;
#define ISG_P3(ACT_, PO_, slot)                                                                                   \
  do {                                                                                                            \
    if (!attr_set[slot][dev]) {                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_h3p_kernel<ACT_, PO_>),                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM) != hipSuccess)                 \
        return ISG_ELAUNCH;                                                                                       \
      attr_set[slot][dev] = true;                                                                                 \
    }                                                                                                             \
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
