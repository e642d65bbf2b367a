// fp32 linear layers on the bf16 matrix cores: D[M,N] = act(A[M,K] . W[N,K]^T + bias)
//
// gfx950 has no TF32/xf32 and its f32-input MFMA runs at the f32 vector rate (157 TF); bf16 MFMA is 16x that.  An fp32
// value splits EXACTLY into three bf16 terms x = x1 + x2 + x3 (8 + 8 + 8 significant bits, each term the bf16 rounding
// of the remainder), and the six products with i + j <= 4
//     x1*y1 + (x1*y2 + x2*y1) + (x2*y2 + x1*y3 + x3*y1)
// drop only terms below 2^-24 of the result: fp32-level accuracy (measured rel. rms 1.2e-7 against fp64, vs 2.9e-7 for a
// plain fp32 GEMM) at 6/16 of the f32-MFMA time.  Each bf16 x bf16 product is exact in the MFMA's fp32 accumulator.
//
//   tile    128 x 128 per workgroup (8 waves, 4 x 2; a wave owns 32 x 64 = 1 x 2 MFMA tiles of 32x32), BK = 32;
//           2 workgroups per CU -> 4 waves per SIMD whose split / LDS / MFMA phases interleave
//   A       fp32 in HBM, loaded as float4, split into its three bf16 planes in registers (v_cvt_pk_bf16_f32 = RNE),
//           written to LDS; W is static, so its planes are split once (isg_split_bf16x3) and streamed as bf16
//   LDS     [plane][row][32 + 8 pad] bf16: 80-byte rows keep the 16-byte fragment reads (ds_read_b128) conflict-free
//   MFMA    v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5)..+7] and W[n = l&31][same k]; 6 per
//           (tile, k-step), small terms first
//   pipeline single LDS buffer + TWO register images: a tile's global loads are issued two k-steps ahead and get two
//           MFMA phases (2 x 24 MFMAs per wave, x 4 waves per SIMD) to land
//   epilogue acc (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) -> + bias -> optional exact GELU -> transposed
//           through a per-wave LDS patch -> 16-byte-per-lane row stores
#include "isg_common.hpp"

#include <stdlib.h>

namespace isg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 32;
constexpr int GM_LD = GM_BK + 8;   // bf16 elements per LDS row (80 bytes)

__device__ __forceinline__ float bf16_to_f32(__bf16 v) {
  return __uint_as_float(((unsigned)__builtin_bit_cast(unsigned short, v)) << 16);
}

// keep a value alive without using it (ablation builds only)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void keep(const float4 &v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ __forceinline__ void keep(const bf16x8 &v) { asm volatile("" ::"v"(__builtin_bit_cast(i32x4, v))); }

// x -> (x1, x2, x3), xk = bf16(remainder)
__device__ __forceinline__ void split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
  p1 = (__bf16)x;
  const float r1 = x - bf16_to_f32(p1);
  p2 = (__bf16)r1;
  const float r2 = r1 - bf16_to_f32(p2);
  p3 = (__bf16)r2;
}

// planes[q][row][kp] (kp = K rounded up to GM_BK, zero filled) from fp32 w[row][K]
__global__ void split_bf16x3_kernel(const float *__restrict__ w, int rows, int K, int Kp, __bf16 *__restrict__ planes) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)rows * Kp;
  if (idx >= total) return;
  const int r = (int)(idx / Kp), k = (int)(idx - (int64_t)r * Kp);
  __bf16 p1 = (__bf16)0.f, p2 = p1, p3 = p1;
  if (k < K) split3(w[(int64_t)r * K + k], p1, p2, p3);
  planes[idx] = p1;
  planes[total + idx] = p2;
  planes[2 * total + idx] = p3;
}

// DBG: compile-time ablation switches for profiling (outputs wrong unless 0): 1 no MFMA, 2 no fragment reads,
// 4 no split / LDS stores, 8 no in-loop global loads, 16 no epilogue stores
// A16 / D16: A is read / D is written as fp16 (feature rows of the fp16 message-passing variant); lda / ldd in elements
typedef _Float16 gm_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned gm_u32x2 __attribute__((ext_vector_type(2)));
template <bool A16> struct GmRawA { typedef float4 type; };
template <> struct GmRawA<true> { typedef gm_u32x2 type; };
__device__ __forceinline__ void gm_zero(float4 &v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void gm_zero(gm_u32x2 &v) { v = gm_u32x2{0u, 0u}; }
__device__ __forceinline__ float4 gm_cvt(const float4 &v) { return v; }
__device__ __forceinline__ float4 gm_cvt(const gm_u32x2 &v) {
  const gm_f16x4 h = __builtin_bit_cast(gm_f16x4, v);
  return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
}

// DUAL (long reductions, K > 256): the five small product terms accumulate in a second accumulator, so that the main
// chain takes ONE rounding at the magnitude of the result per 16-deep k-step instead of six -- a K-long sequential fp32
// chain otherwise carries ~sqrt(6 K / 16) half-ulps of error (measured 4.2x a blocked CPU GEMM's at K = 1200, 4.4x at
// K = 2048; 1.7x / 1.8x with the second accumulator).  Its 32 registers are paid for with one register image instead of
// two (loads one k-tile ahead), which keeps 4 waves per SIMD.
template <int ACT, int DBG, bool A16, bool D16, bool XCD, bool DUAL = false>   // ACT: 0 none, 1 exact GELU, 2 ReLU; XCD: tile order, see below
__global__ __launch_bounds__(512, 4) void linear_bf16x6_kernel(const float *__restrict__ A, const __bf16 *__restrict__ Wp,
                                                            const float *__restrict__ bias, float *__restrict__ D, int M,
                                                            int N, int K, int Kp, int lda, int ldd, int nt_store) {
  typedef typename GmRawA<A16>::type RawA;
  const void *Av = A;   // fp16 variants: the same pointers, half elements
  void *Dv = D;
  // one LDS object, so that the epilogue's per-wave patches (36.9 KB) may legally run from the A planes into the W planes
  struct Smem {
    __bf16 a[3][GM_BM][GM_LD];
    __bf16 b[3][GM_BN][GM_LD];
  };
  __shared__ __attribute__((aligned(16))) Smem sm;
#define sA sm.a
#define sB sm.b
  static_assert(sizeof(Smem) >= 8 * 32 * 36 * sizeof(float), "the epilogue patches must fit the operand block");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2.  The n-tiles
  // of one row-block read the same A rows, so they are made consecutive WITHIN an XCD (ids congruent mod 8): the A tile
  // is fetched into one L2 once instead of into up to min(8, n-tiles) of them.  gridDim.y is padded to a multiple of 8.
  int n0, m0;
  if (XCD) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, nt = gridDim.x;
    const int slot = L >> 3;
    const int mt = (L & 7) + 8 * (slot / nt);
    n0 = (slot % nt) * GM_BN;
    m0 = mt * GM_BM;
    if (m0 >= M) return;
  } else {
    n0 = blockIdx.x * GM_BN;
    m0 = blockIdx.y * GM_BM;
  }
  const int64_t plane_stride = (int64_t)N * Kp;

  f32x16 acc[1][2], acc2[DUAL ? 2 : 1];
#pragma unroll
  for (int i = 0; i < 1; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (DUAL) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
  }

  // two register images (tiles t+1 and t+2): a tile's global loads get two MFMA phases to land
  RawA ra0[2], ra1[2];          // A: 4 elements per thread per tile and u, raw (fp32: 16 B, fp16: 8 B)
  bf16x8 rb0[3], rb1[3];        // W: 16 B per plane per thread per tile

  // next tile -> registers (global loads only; the LDS image is written below, in one place, so that every LDS access
  // stays a direct __shared__ access)
#define GM_LOAD_TILE(RA, RB, k0)                                                                                    \
  {                                                                                                            \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                            \
      const int i = tid + 512 * u;                                                                             \
      const int row = i >> 3, c4 = i & 7;                                                                      \
      /* never a conditional load (the compiler would wait for it at once): clamp the address, mask at store time */ \
      const int gr = min(m0 + row, M - 1), gk = min((k0) + c4 * 4, K - 4);                                     \
      if constexpr (A16)                                                                                       \
        RA[u] = *reinterpret_cast<const RawA *>(reinterpret_cast<const _Float16 *>(Av) + (int64_t)gr * lda + gk); \
      else                                                                                                     \
        RA[u] = *reinterpret_cast<const RawA *>(A + (int64_t)gr * lda + gk);                                   \
    }                                                                                                          \
    {                                                                                                          \
      const int row = tid >> 2, c8 = tid & 3;                                                                  \
      const int gn = min(n0 + row, N - 1);   /* rows past N repeat the last row; their columns are never stored */ \
      _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                            \
        RB[q] = *reinterpret_cast<const bf16x8 *>(Wp + q * plane_stride + (int64_t)gn * Kp + (k0) + c8 * 8);   \
    }                                                                                                          \
  }

  const int nk = Kp / GM_BK;
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  // Optional (ISG_GEMM_KROT=1, flags bit 1; OFF by default): row-blocks walk K from different starting tiles (wrapping), so
  // that workgroups do not touch the same 128-byte column of their rows at the same moment ("partition camping").  It makes
  // the ORDER of a row's fp32 accumulation depend on where the row sits in the batch: the same graph then gives
  // bit-different projections once it is re-batched or sharded, which breaks "a shard's result is the path run on that
  // shard alone" (distributed.py) and can flip a near-tie in the top-k mask.  Default: every row-block starts at tile 0.
  const int kshift = (nt_store & 2) ? (int)((unsigned)(m0 / GM_BM) % (unsigned)nk) : 0;
#define GM_KT(t) ((((t) + kshift) >= nk ? (t) + kshift - nk : (t) + kshift))
#define GM_STEP(RA, RB, kt)                                                                                    \
  {                                                                                                            \
    if ((kt) > 0) __syncthreads(); /* everyone is done reading the previous tile */                            \
    /* registers -> LDS: split the fp32 A values into their three bf16 planes, copy the W planes */            \
    if constexpr (DBG & 4) {                                                                                   \
      keep(gm_cvt(RA[0]));                                                                                     \
      keep(gm_cvt(RA[1]));                                                                                     \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) keep(RB[q]);                                               \
    } else _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                     \
      const int i = tid + 512 * u;                                                                             \
      const int row = i >> 3, c4 = i & 7;                                                                      \
      if (m0 + row >= M || GM_KT(kt) * GM_BK + c4 * 4 >= K) gm_zero(RA[u]);                                    \
      const float4 av = gm_cvt(RA[u]);                                                                         \
      bf16x4 p0, p1, p2;                                                                                       \
      __bf16 t0, t1, t2;                                                                                       \
      split3(av.x, t0, t1, t2); p0[0] = t0; p1[0] = t1; p2[0] = t2;                                            \
      split3(av.y, t0, t1, t2); p0[1] = t0; p1[1] = t1; p2[1] = t2;                                            \
      split3(av.z, t0, t1, t2); p0[2] = t0; p1[2] = t1; p2[2] = t2;                                            \
      split3(av.w, t0, t1, t2); p0[3] = t0; p1[3] = t1; p2[3] = t2;                                            \
      *reinterpret_cast<bf16x4 *>(&sA[0][row][c4 * 4]) = p0;                                                   \
      *reinterpret_cast<bf16x4 *>(&sA[1][row][c4 * 4]) = p1;                                                   \
      *reinterpret_cast<bf16x4 *>(&sA[2][row][c4 * 4]) = p2;                                                   \
    }                                                                                                          \
    if constexpr (!(DBG & 4)) {                                                                                \
      const int row = tid >> 2, c8 = tid & 3;                                                                  \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8 *>(&sB[q][row][c8 * 8]) = RB[q];  \
    }                                                                                                          \
    __syncthreads();                                                                                           \
    if constexpr (!(DBG & 8))                                                                                  \
      if ((kt) + (DUAL ? 1 : 2) < nk) GM_LOAD_TILE(RA, RB, GM_KT((kt) + (DUAL ? 1 : 2)) * GM_BK) /* lands two (DUAL: one) MFMA phases from now */ \
    if constexpr (!(DBG & 2)) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                               \
      bf16x8 a[3], b[2][3];                                                                                    \
      _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                            \
        a[q] = *reinterpret_cast<const bf16x8 *>(&sA[q][wm * 32 + fr][ks * 16 + fk]);                          \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                          \
          b[j][q] = *reinterpret_cast<const bf16x8 *>(&sB[q][wn * 64 + j * 32 + fr][ks * 16 + fk]);            \
      if constexpr (DBG & 1) {                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                        \
          keep(a[q]);                                                                                          \
          keep(b[0][q]);                                                                                       \
          keep(b[1][q]);                                                                                       \
        }                                                                                                      \
      } else if constexpr (DUAL) {                                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
          f32x16 c2 = acc2[j];                                                                                 \
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][2], c2, 0, 0, 0);                            \
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[j][0], c2, 0, 0, 0);                            \
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][1], c2, 0, 0, 0);                            \
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][1], c2, 0, 0, 0);                            \
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][0], c2, 0, 0, 0);                            \
          acc2[j] = c2;                                                                                        \
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][0], acc[0][j], 0, 0, 0);              \
        }                                                                                                      \
      } else _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
        f32x16 c = acc[0][j];                                                                                  \
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][2], c, 0, 0, 0);                                \
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[j][0], c, 0, 0, 0);                                \
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][1], c, 0, 0, 0);                                \
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][1], c, 0, 0, 0);                                \
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][0], c, 0, 0, 0);                                \
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][0], c, 0, 0, 0);                                \
        acc[0][j] = c;                                                                                         \
      }                                                                                                        \
    }                                                                                                          \
  }

  GM_LOAD_TILE(ra0, rb0, GM_KT(0) * GM_BK)
  if constexpr (DUAL) {
    for (int kt = 0; kt < nk; ++kt) GM_STEP(ra0, rb0, kt)
  } else {
    if (nk > 1) GM_LOAD_TILE(ra1, rb1, GM_KT(1) * GM_BK)
    for (int kt = 0; kt < nk; kt += 2) {
      GM_STEP(ra0, rb0, kt)
      if (kt + 1 < nk) GM_STEP(ra1, rb1, kt + 1)
    }
  }
#undef GM_STEP
#undef GM_LOAD_TILE
#undef GM_KT

  // ---- epilogue: + bias, optional GELU, then each 32x32 accumulator tile goes through a private LDS patch so that the
  //      global stores are 16 bytes per lane along a row (4 store instructions per tile instead of 16 dword stores) ----
  __syncthreads();   // every wave is done with the operand tiles: the LDS is free
  float *patch = reinterpret_cast<float *>(&sm) + wave * (32 * 36);   // [32][32 + 4 pad] floats per wave
  const int h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int colb = n0 + wn * 64 + j * 32;
    const int col = colb + fr;
    const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[0][j][r];
      if constexpr (DUAL) v += acc2[j][r];
      v += bv;
      if (ACT == 1) v = gelu_exact(v);
      if (ACT == 2) v = v < 0.f ? 0.f : v;   // ReLU (a NaN stays a NaN, as in torch)
      patch[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + fr] = v;
    }
    __builtin_amdgcn_wave_barrier();
    const int rowb = m0 + wm * 32;
    const bool vec_ok = (ldd & 3) == 0 && colb + 32 <= N && (reinterpret_cast<uintptr_t>(Dv) & 15) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = q * 64 + lane;
      const int rr = idx >> 3, c4 = idx & 7;
      const float4 v = *reinterpret_cast<const float4 *>(&patch[rr * 36 + c4 * 4]);
      const int row = rowb + rr;
      if constexpr (D16) {
        if ((DBG & 16) ? (row < 0) : (row < M)) {
          _Float16 *dst = reinterpret_cast<_Float16 *>(Dv) + (int64_t)row * ldd + colb + c4 * 4;
          if (vec_ok) {
            gm_f16x4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            *reinterpret_cast<gm_f16x4 *>(dst) = hv;
          } else {
            if (colb + c4 * 4 + 0 < N) dst[0] = (_Float16)v.x;
            if (colb + c4 * 4 + 1 < N) dst[1] = (_Float16)v.y;
            if (colb + c4 * 4 + 2 < N) dst[2] = (_Float16)v.z;
            if (colb + c4 * 4 + 3 < N) dst[3] = (_Float16)v.w;
          }
        }
      } else if ((DBG & 16) ? (row < 0) : (row < M)) {
        float *dst = D + (int64_t)row * ldd + colb + c4 * 4;
        if (vec_ok) {
          if (nt_store & 1) {   // streaming result (see isg_linear launch): do not displace the operands in L2 / Infinity Cache
            typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
            nt_f32x4 w4 = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w4, reinterpret_cast<nt_f32x4 *>(dst));
          } else {
            *reinterpret_cast<float4 *>(dst) = v;
          }
        } else {
          if (colb + c4 * 4 + 0 < N) dst[0] = v.x;
          if (colb + c4 * 4 + 1 < N) dst[1] = v.y;
          if (colb + c4 * 4 + 2 < N) dst[2] = v.z;
          if (colb + c4 * 4 + 3 < N) dst[3] = v.w;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#undef sA
#undef sB
}

}  // namespace isg

using namespace isg;

extern "C" int isg_split_bf16x3(const float *w, int64_t rows, int32_t K, uint16_t *planes, void *stream) {
  if (rows < 0 || K <= 0) return ISG_EINVAL;
  if (rows == 0) return ISG_OK;
  if (!w || !planes) return ISG_EINVAL;
  const int Kp = (K + GM_BK - 1) / GM_BK * GM_BK;
  const int64_t total = rows * Kp;
  if (rows >= (1ll << 31) || (total + 255) / 256 >= (1ll << 31)) return ISG_EUNSUPPORTED;
  split_bf16x3_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(w, (int)rows, K, Kp,
                                                                                      reinterpret_cast<__bf16 *>(planes));
  return check_launch();
}

static int linear_launch(const void *a, int a16, const uint16_t *w_planes, const float *bias, void *d, int d16,
                         int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || lda < K || ldd < N || act < 0 || act > 2) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a || !w_planes || !d) return ISG_EINVAL;
  // 4-element loads of A need aligned rows (16 bytes fp32, 8 bytes fp16) and K a multiple of 4
  const uintptr_t amask = a16 ? 7 : 15;
  if ((K & 3) != 0 || (lda & 3) != 0 || (reinterpret_cast<uintptr_t>(a) & amask) != 0 || M >= (1ll << 31)) return ISG_EUNSUPPORTED;
  const int Kp = (K + GM_BK - 1) / GM_BK * GM_BK;
  const long long mt = (M + GM_BM - 1) / GM_BM;
  if (mt > 65535) return ISG_EUNSUPPORTED;
  dim3 grid((unsigned)((N + GM_BN - 1) / GM_BN), (unsigned)mt), block(512);
  const __bf16 *wp = reinterpret_cast<const __bf16 *>(w_planes);
  hipStream_t st = as_stream(stream);
  // results of at least ISG_GEMM_NT_MB MB (default 128) are written with non-temporal stores: the big projected rows
  // (x_l|x_r 337 MB, e_proj 420 MB) exceed the caches and are consumed by a LATER kernel -- which itself runs 6 % faster
  // when they were streamed -- while the operands (A re-read per n-tile, W planes) and the small results that the very
  // next kernel reads (x_proj, MLPs: <= 84 MB) should stay resident (profiles/r01_e, r01_q / r01_x)
  static const long long nt_mb = [] { const char *e = getenv("ISG_GEMM_NT_MB"); return e ? atoll(e) : 128ll; }();   // read once
  static const int krot = [] { const char *e = getenv("ISG_GEMM_KROT"); return e && atoi(e) != 0 ? 2 : 0; }();
  const int nt = (nt_mb >= 0 && (long long)M * N * (d16 ? 2 : 4) >= nt_mb * 1000000ll ? 1 : 0) | krot;   // flags: bit 0 streaming stores, bit 1 K rotation
  dim3 gridx(grid.x, (grid.y + 7) / 8 * 8);
#ifdef ISG_GEMM_ABLATION   // profiling build only (tools/build_ablation.sh): select a compile-time ablated variant by env
  const char *dv = getenv("ISG_GEMM_DBG");
  const int dbg = dv ? atoi(dv) : 0;
#define ISG_DBG_CASE(v) if (dbg == v) { linear_bf16x6_kernel<0, v, false, false, false><<<grid, block, 0, st>>>((const float *)a, wp, bias, (float *)d, (int)M, N, K, Kp, lda, ldd, nt); return check_launch(); }
  ISG_DBG_CASE(1) ISG_DBG_CASE(3) ISG_DBG_CASE(7) ISG_DBG_CASE(15) ISG_DBG_CASE(16) ISG_DBG_CASE(31) ISG_DBG_CASE(8) ISG_DBG_CASE(4)
#undef ISG_DBG_CASE
#endif
  static const bool xcd_on = getenv("ISG_GEMM_NO_XCD") == nullptr;   // read once
  const bool xcd = grid.x > 1 && xcd_on;
  static const int dual_k = [] { const char *e = getenv("ISG_GEMM_DUAL_K"); return e ? atoi(e) : 256; }();   // read once
  const bool dual = K > dual_k;
#define ISG_LIN(ACT_, A_, D_)                                                                                                \
  do {                                                                                                                       \
    if (dual) {                                                                                                              \
      if (xcd) linear_bf16x6_kernel<ACT_, 0, A_, D_, true, true><<<gridx, block, 0, st>>>((const float *)a, wp, bias, (float *)d, (int)M, N, K, Kp, lda, ldd, nt); \
      else linear_bf16x6_kernel<ACT_, 0, A_, D_, false, true><<<grid, block, 0, st>>>((const float *)a, wp, bias, (float *)d, (int)M, N, K, Kp, lda, ldd, nt); \
    } else if (xcd) linear_bf16x6_kernel<ACT_, 0, A_, D_, true><<<gridx, block, 0, st>>>((const float *)a, wp, bias, (float *)d, (int)M, N, K, Kp, lda, ldd, nt); \
    else linear_bf16x6_kernel<ACT_, 0, A_, D_, false><<<grid, block, 0, st>>>((const float *)a, wp, bias, (float *)d, (int)M, N, K, Kp, lda, ldd, nt); \
  } while (0)
  if (!a16 && !d16) { if (act == 1) ISG_LIN(1, false, false); else if (act == 2) ISG_LIN(2, false, false); else ISG_LIN(0, false, false); }
  else if (act == 2) return ISG_EUNSUPPORTED;   // ReLU exists for the fp32 text encoder only
  else if (a16 && !d16) { if (act == 1) ISG_LIN(1, true, false); else ISG_LIN(0, true, false); }
  else if (!a16 && d16) { if (act == 1) ISG_LIN(1, false, true); else ISG_LIN(0, false, true); }
  else { if (act == 1) ISG_LIN(1, true, true); else ISG_LIN(0, true, true); }
#undef ISG_LIN
  return check_launch();
}

extern "C" int isg_linear_bf16x6(const float *a, const uint16_t *w_planes, const float *bias, float *d, int64_t M,
                                 int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act, void *stream) {
  return linear_launch(a, 0, w_planes, bias, d, 0, M, N, K, lda, ldd, act, stream);
}

extern "C" int isg_linear_bf16x6_f16(const void *a, int32_t a_is_f16, const uint16_t *w_planes, const float *bias, void *d,
                                     int32_t d_is_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd,
                                     int32_t act, void *stream) {
  return linear_launch(a, a_is_f16 != 0, w_planes, bias, d, d_is_f16 != 0, M, N, K, lda, ldd, act, stream);
}
