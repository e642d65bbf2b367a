// fp32 linear layers on the fp16 matrix cores with THREE products per term: D[M,N] = act(A[M,K] . W[N,K]^T + bias), K <= 128
//
// isg_gemm_panel.hip splits every fp32 operand into three bf16 planes (8 significant bits each) and needs six MFMA
// products for fp32-level accuracy.  An fp16 plane carries 11 bits: x * s = hi + mid + r with |r| <= 2^-24 |x * s| (two
// round-to-nearest fp16 terms of the exactly scaled value), so the three products
//     hi_a hi_b + (hi_a mid_b + mid_a hi_b)
// drop only mid_a mid_b (<= 2^-24 of the product) and the operand residues (2^-24 each): the error of ONE extra fp32
// rounding per operand, below the rounding of the fp32 accumulation itself.  Measured: 1.0-1.1x the error of a plain fp32
// GEMM (bf16x6: 0.6x) at HALF the matrix-core work of bf16x6, two LDS planes instead of three and two W planes from L2.
// What fp16 lacks is range: `mid` ~ 2^-12 |x| falls into the subnormals (absolute spacing 2^-24) for |x| < 2^-2.  So
// every A row and every W row (output column) is scaled by its own power of two that puts its largest magnitude into
// [2^13, 2^14) -- exact -- and the accumulator is scaled back in the epilogue (two exact multiplies).  A row's scale needs
// the whole row, which is why this form exists for K <= 128 only (the row panel is staged in one piece: lin_edge,
// lin_l | lin_r); everything else about the kernel is isg_gemm_panel.hip's A-stationary panel.
#include "isg_f16x3.hpp"

#include <stdlib.h>

namespace isg {

constexpr int H3_BM = 64, H3_KC = 128, H3_LD = H3_KC + 8, H3_THREADS = 256;

// Wf[q][nt][ks][lane][j] (q = hi, mid) of w[n][k] * s_n, n = nt*32 + (lane & 31), k = ks*16 + 8*(lane >> 5) + j; inv[n] = 1/s_n
__global__ void split_f16x2_frag_kernel(const float *__restrict__ w, int N, int K, int NT, int KS, _Float16 *__restrict__ out,
                                        float *__restrict__ inv_scale) {
  const int n = blockIdx.x;            // one workgroup per weight row (padded rows: zeros, scale 1)
  __shared__ float s_s;
  float mx = 0.f;
  if (n < N)
    for (int k = threadIdx.x; k < K; k += blockDim.x) mx = fmaxf(mx, fabsf(w[(int64_t)n * K + k]));
  mx = wave_max(mx);
  if (threadIdx.x == 0) {
    float s, inv;
    h3_scale(mx, s, inv);
    s_s = s;
    inv_scale[n] = inv;
  }
  __syncthreads();
  const float s = s_s;
  const int nt = n >> 5, r = n & 31;
  const int64_t total = (int64_t)NT * KS * 512;
  for (int k = threadIdx.x; k < KS * 16; k += blockDim.x) {
    const int ks = k >> 4, kk = k & 15;
    const int lane = r + 32 * (kk >> 3), j = kk & 7;
    float v = (n < N && k < K) ? w[(int64_t)n * K + k] * s : 0.f;
    const _Float16 hi = (_Float16)v;
    const _Float16 mid = (_Float16)(v - (float)hi);
    const int64_t idx = (((int64_t)nt * KS + ks) * 64 + lane) * 8 + j;
    out[idx] = hi;
    out[total + idx] = mid;
  }
}

// F16D: the result as half rows (one RNE rounding of the fp32 value: BASELINE configs[4]'s "fp16 features / fp32 accumulate")
template <int ACT, int WTN, bool F16D = false>
__global__ __launch_bounds__(H3_THREADS, 2) void linear_f16x3_kernel(const float *__restrict__ A,
                                                                     const _Float16 *__restrict__ Wf,
                                                                     const float *__restrict__ w_inv,
                                                                     const float *__restrict__ bias, void *__restrict__ Dv,
                                                                     int M, int N, int K, int KS, int NT, int lda, int ldd,
                                                                     int nt_store, int out_cols, int64_t out_stride) {
  __shared__ __attribute__((aligned(16))) _Float16 sA[2][H3_BM][H3_LD];   // 34,816 B
  __shared__ float s_inv[H3_BM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * H3_BM;
  const int npass = (NT + 4 * WTN - 1) / (4 * WTN);
  const int fr = lane & 31, fk = (lane >> 5) * 8;

  // ---- stage the panel once: global -> registers -> row scale -> (hi, mid) planes -----------------------------------
  {
    float4 ra[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + H3_THREADS * u;
      const int row = i >> 5, c4 = i & 31;
      const int gr = min(m0 + row, M - 1), gk = min(c4 * 4, K - 4);
      ra[u] = *reinterpret_cast<const float4 *>(A + (int64_t)gr * lda + gk);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + H3_THREADS * u;
      const int row = i >> 5, c4 = i & 31;       // the 32 lanes of a half-wave hold one row
      float4 v = ra[u];
      if (m0 + row >= M || c4 * 4 >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
      const float mx = group_max<32>(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      float s, inv;
      h3_scale(mx, s, inv);
      if (c4 == 0) s_inv[row] = inv;
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      hf16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
      hf16x4 mid = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                    (_Float16)(v.w - (float)hi[3])};
      *reinterpret_cast<hf16x4 *>(&sA[0][row][c4 * 4]) = hi;
      *reinterpret_cast<hf16x4 *>(&sA[1][row][c4 * 4]) = mid;
    }
  }
  __syncthreads();

  const unsigned plane_b = (unsigned)NT * (unsigned)KS * 1024u;      // bytes per W plane
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(Wf), 0, (int)(2u * plane_b), 0x00020000);
  const int voff = lane * 16;

#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    const int nt0 = (pass * 4 + wave) * WTN;
    const bool active = nt0 < NT;
    hf32x16 acc[2][WTN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    unsigned wb[WTN];
    float bv[WTN], wi[WTN];     // bias and inverse W scale of this lane's column, loaded ahead of the k loop
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      wb[j] = (unsigned)min(nt0 + j, NT - 1) * (unsigned)KS * 1024u;
      const int col = min((nt0 + j) * 32 + fr, N - 1);
      bv[j] = bias ? bias[col] : 0.f;
      wi[j] = w_inv[col];
    }
    hf16x8 a0[2][2], a1[2][2], b0[WTN][2], b1[WTN][2];
#define H3_LOAD_B(B, s)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < WTN; ++j) _Pragma("unroll") for (int q = 0; q < 2; ++q)                  \
      B[j][q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(                                \
          wrsrc, voff, (int)(wb[j] + q * plane_b + (unsigned)(s) * 1024u), 0));
#define H3_LOAD_A(Afr, ksl)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q)                    \
      Afr[i][q] = *reinterpret_cast<const hf16x8 *>(&sA[q][i * 32 + fr][(ksl) * 16 + fk]);
#define H3_MMA(Afr, B)                                                                                           \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < WTN; ++j) {                \
    hf32x16 c = acc[i][j];                                                                                       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Afr[i][0], B[j][1], c, 0, 0, 0);                                  \
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Afr[i][1], B[j][0], c, 0, 0, 0);                                  \
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Afr[i][0], B[j][0], c, 0, 0, 0);                                  \
    acc[i][j] = c;                                                                                               \
  }
    if (active) {
      H3_LOAD_B(b0, 0)
      H3_LOAD_A(a0, 0)
      int ks = 0;
#pragma unroll 1
      for (; ks + 2 <= KS; ks += 2) {
        H3_LOAD_B(b1, ks + 1)
        H3_LOAD_A(a1, ks + 1)
        H3_MMA(a0, b0)
        H3_LOAD_B(b0, min(ks + 2, KS - 1))
        H3_LOAD_A(a0, min(ks + 2, 7))
        H3_MMA(a1, b1)
      }
      if (ks < KS) H3_MMA(a0, b0)
    }
#undef H3_LOAD_B
#undef H3_LOAD_A
#undef H3_MMA
    if (active) {
      const int h = lane >> 5;
      const bool rows_full = m0 + H3_BM <= M;
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        if (nt0 + j >= NT) break;
        const int col = (nt0 + j) * 32 + fr;
        const bool cols_full = (nt0 + j) * 32 + 32 <= N;
        const int otile = ((nt0 + j) * 32) / out_cols;
        const int64_t obase = (int64_t)otile * out_stride - (int64_t)otile * out_cols;
#define H3_EPI(GUARD)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) {                 \
    const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;                                                      \
    const int row = m0 + rl;                                                                                     \
    float v = (acc[i][j][r] * s_inv[rl]) * wi[j] + bv[j];     /* both scales are powers of two: exact */        \
    if (ACT == 1) v = gelu_exact(v);                                                                             \
    if (GUARD) {                                                                                                 \
      if (F16D) {     /* the fp32 value first, THEN one rounding: the compiler would fold the last fma into a v_fma_mix */ \
        asm volatile("" : "+v"(v));     /* (single rounding of the exact sum: differs from fp32-then-half on ties)        */ \
        _Float16 *dst = reinterpret_cast<_Float16 *>(Dv) + obase + (int64_t)row * ldd + col;                     \
        if (nt_store) __builtin_nontemporal_store((_Float16)v, dst);                                             \
        else *dst = (_Float16)v;                                                                                 \
      } else {                                                                                                   \
        float *dst = reinterpret_cast<float *>(Dv) + obase + (int64_t)row * ldd + col;                           \
        if (nt_store) __builtin_nontemporal_store(v, dst);                                                       \
        else *dst = v;                                                                                           \
      }                                                                                                          \
    }                                                                                                            \
  }
        if (rows_full && cols_full) H3_EPI(true)
        else H3_EPI(col < N && row < M)
#undef H3_EPI
      }
    }
  }
}

}  // namespace isg

using namespace isg;

extern "C" int64_t isg_split_f16x2_frag_elems(int64_t rows, int32_t K) {
  if (rows <= 0 || K <= 0) return 0;
  return 2ll * ((rows + 31) / 32) * ((K + 15) / 16) * 512;
}

// planes: uint16[isg_split_f16x2_frag_elems]; inv_scale: float[ceil(rows / 32) * 32]
extern "C" int isg_split_f16x2_frag(const float *w, int64_t rows, int32_t K, uint16_t *planes, float *inv_scale,
                                    void *stream) {
  if (rows < 0 || K <= 0) return ISG_EINVAL;
  if (rows == 0) return ISG_OK;
  if (!w || !planes || !inv_scale) return ISG_EINVAL;
  if (rows >= (1ll << 24)) return ISG_EUNSUPPORTED;
  const int NT = (int)((rows + 31) / 32), KS = (K + 15) / 16;
  split_f16x2_frag_kernel<<<(unsigned)(NT * 32), 64, 0, as_stream(stream)>>>(w, (int)rows, K, NT, KS,
                                                                             reinterpret_cast<_Float16 *>(planes), inv_scale);
  return check_launch();
}

static int linear_f16x3_launch(const float *a, const uint16_t *w_frag, const float *w_inv_scale, const float *bias, void *d,
                               bool d_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                               int32_t out_cols, int64_t out_stride, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || lda < K || act < 0 || act > 1) return ISG_EINVAL;
  if (out_cols <= 0 || ldd < out_cols || (out_cols < N && (out_cols & 31) != 0) || N % out_cols != 0) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a || !w_frag || !w_inv_scale || !d) return ISG_EINVAL;
  if (K > H3_KC || (K & 3) != 0 || (lda & 3) != 0 || (reinterpret_cast<uintptr_t>(a) & 15) != 0) return ISG_EUNSUPPORTED;
  const long long panels = (M + H3_BM - 1) / H3_BM;
  if (panels >= (1ll << 31) || M >= (1ll << 31)) return ISG_EUNSUPPORTED;
  const int NT = (N + 31) / 32, KS = (K + 15) / 16;
  auto waste = [&](int w) { const int per = 4 * w; return ((NT + per - 1) / per) * per - NT; };
  const int wtn = waste(2) <= waste(1) + 1 ? 2 : 1;
  static const long long nt_b = [] { const char *e = getenv("ISG_GEMM_NT_MB"); return (e ? atoll(e) : 128) * 1000000ll; }();
  const int nt = nt_b >= 0 && (long long)M * N * (d_f16 ? 2 : 4) >= nt_b;
  const _Float16 *wf = reinterpret_cast<const _Float16 *>(w_frag);
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)panels), block(H3_THREADS);
#define ISG_H3(ACT_, W_, F_) linear_f16x3_kernel<ACT_, W_, F_><<<grid, block, 0, st>>>(a, wf, w_inv_scale, bias, d, (int)M, N, K, KS, NT, lda, ldd, nt, out_cols, (long long)out_stride)
  if (d_f16) {
    if (act == 1) { if (wtn == 2) ISG_H3(1, 2, true); else ISG_H3(1, 1, true); }
    else { if (wtn == 2) ISG_H3(0, 2, true); else ISG_H3(0, 1, true); }
  } else {
    if (act == 1) { if (wtn == 2) ISG_H3(1, 2, false); else ISG_H3(1, 1, false); }
    else { if (wtn == 2) ISG_H3(0, 2, false); else ISG_H3(0, 1, false); }
  }
#undef ISG_H3
  return check_launch();
}

extern "C" int isg_linear_f16x3(const float *a, const uint16_t *w_frag, const float *w_inv_scale, const float *bias,
                                float *d, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                                int32_t out_cols, int64_t out_stride, void *stream) {
  return linear_f16x3_launch(a, w_frag, w_inv_scale, bias, d, false, M, N, K, lda, ldd, act, out_cols, out_stride, stream);
}

// The same product with the result rounded ONCE to half rows (ldd / out_stride in halves): the projections of a model in
// BASELINE configs[4]'s "fp16 features / fp32 accumulate" mode (x_l | x_r, e_proj), at half the matrix-core work of
// isg_linear_panel's bf16 six-product form.
extern "C" int isg_linear_f16x3_f16(const float *a, const uint16_t *w_frag, const float *w_inv_scale, const float *bias,
                                    uint16_t *d, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                                    int32_t out_cols, int64_t out_stride, void *stream) {
  return linear_f16x3_launch(a, w_frag, w_inv_scale, bias, d, true, M, N, K, lda, ldd, act, out_cols, out_stride, stream);
}

// =====================================================================================================================
// Tile form for K > 128 (x_proj: 512 -> 256 -> 128 at configs[1]).  A row's scale needs the row's largest magnitude over
// ALL of K before the first k-tile is split, so it is an INPUT: a_rowmax[M, P], P partial maxima per row, written by the
// kernel that produced A (the message-passing kernel per head, this kernel's own epilogue per 32-column subtile) -- a few
// KB next to the rows, no extra pass over A.  Structure of isg_gemm.hip's tile kernel (128 x 128 x 32 tile, 8 waves as
// 4 x 2, one LDS buffer, two register images, XCD-aware tile order, results through per-wave LDS patches as 16-byte row
// stores), two planes instead of three, three MFMAs per product instead of six.
// =====================================================================================================================
namespace isg {

constexpr int HT_BM = 128, HT_BN = 128, HT_BK = 32, HT_LD = HT_BK + 8;

// planes[q][row][Kp] (q = hi, mid; Kp = K rounded up to 32, zero padded) of w[row][k] * s_row; inv[row] = 1 / s_row
__global__ void split_f16x2_rows_kernel(const float *__restrict__ w, int N, int K, int Kp, _Float16 *__restrict__ planes,
                                        float *__restrict__ inv_scale) {
  const int n = blockIdx.x;
  __shared__ float s_s;
  float mx = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) mx = fmaxf(mx, fabsf(w[(int64_t)n * K + k]));
  mx = wave_max(mx);
  if (threadIdx.x == 0) {
    float s, inv;
    h3_scale(mx, s, inv);
    s_s = s;
    inv_scale[n] = inv;
  }
  __syncthreads();
  const float s = s_s;
  const int64_t total = (int64_t)N * Kp;
  for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
    const float v = k < K ? w[(int64_t)n * K + k] * s : 0.f;
    const _Float16 hi = (_Float16)v;
    planes[(int64_t)n * Kp + k] = hi;
    planes[total + (int64_t)n * Kp + k] = (_Float16)(v - (float)hi);
  }
}

// ACCUM: D already holds the sum over an earlier part of K (reductions longer than 1024 are run as K-chunks, each its own
// accumulation chain, so the fp32 chain error does not grow with K): v = this chunk + D, then bias / activation
// MF: the MFMA shape.  32 = v_mfma_f32_32x32x16_f16 (2 accumulator tiles per wave, two k-halves per k-tile);
// 16 = v_mfma_f32_16x16x32_f16 (8 accumulator tiles of 4 registers, one k-step per k-tile): the same FLOPs, LDS bytes and
// cycles, but the chip holds a higher clock on the 16x16 shape under a dense MFMA stream (MI355X_MICROARCH.md, DVFS item
// 7).  Its fragment reads (row = lane & 15, k = 8 * (lane >> 4)) want a 96-byte row pitch to stay conflict-free.
template <int ACT, bool XCD, bool RMAX, bool ACCUM, int MF>
__global__ __launch_bounds__(512, 4) void linear_f16x3_tile_kernel(const float *__restrict__ A,
                                                                   const float *__restrict__ a_rowmax, int P, int ldp,
                                                                   const _Float16 *__restrict__ Wp,
                                                                   const float *__restrict__ w_inv,
                                                                   const float *__restrict__ bias, float *__restrict__ D,
                                                                   float *__restrict__ d_rowmax, int M, int N, int K,
                                                                   int Kp, int ldw, int lda, int ldd, int nt_store) {
  constexpr int LD = MF == 16 ? HT_BK + 16 : HT_LD;
  struct Smem {
    _Float16 a[2][HT_BM][LD];
    _Float16 b[2][HT_BN][LD];
    float inv[HT_BM];
  };
  __shared__ __attribute__((aligned(16))) Smem sm;
  static_assert(sizeof(_Float16) * 2 * (HT_BM + HT_BN) * LD >= 8 * 32 * 36 * sizeof(float), "epilogue patches must fit");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  int n0, m0;
  if (XCD) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, nt = gridDim.x;
    const int slot = L >> 3;
    const int mt = (L & 7) + 8 * (slot / nt);
    n0 = (slot % nt) * HT_BN;
    m0 = mt * HT_BM;
    if (m0 >= M) return;
  } else {
    n0 = blockIdx.x * HT_BN;
    m0 = blockIdx.y * HT_BM;
  }
  const int64_t plane_stride = (int64_t)N * ldw;     // ldw: row stride of the W planes (the whole weight's padded K)

  // this thread stages the same two rows in every k-tile: their scales once, from the producer's partial maxima
  float sa[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = (tid + 512 * u) >> 3;
    const int gr = min(m0 + row, M - 1);
    float mx = 0.f;
    for (int p = 0; p < P; ++p) mx = fmaxf(mx, a_rowmax[(int64_t)gr * ldp + p]);
    float inv;
    h3_scale(mx, sa[u], inv);
    if (((tid + 512 * u) & 7) == 0) sm.inv[row] = inv;
  }

  hf32x16 acc[2];
  hf32x4 acc16[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc16[i][j] = hf32x4{0.f, 0.f, 0.f, 0.f};
  float4 ra0[2], ra1[2];
  hf16x8 rb0[2], rb1[2];
  // ISG_F16X3_APLANES (timing diagnostic build only, tools/time_f16x3_tile.py with ISG_TOOL_LIB): the bytes of A are read
  // as if they were two fp16 planes [2][M][K] left by A's producer (same bytes, same rows per tile), staged with 16-byte LDS
  // stores, no conversion in the loop: the upper bound of handing activations over pre-split.  The results are garbage.
#ifdef ISG_F16X3_APLANES
  constexpr bool APL = true;
  // ISG_F16X3_APLANES=2: planes interleaved per k-tile, [M][K/32][hi 32 | mid 32] -- a row's k-tile is ONE 128-byte line (with
  // two separate planes it is a 64-byte segment in each, twice the requests on the streamed operand)
  constexpr bool APL_INTERLEAVED = ISG_F16X3_APLANES == 2;
  // ISG_F16X3_APLANES=3: interleaved planes AND the shipped kernel's thread -> address map (8 lanes per row, one 128-byte
  // line per row and k-tile, one instruction): global traffic identical to the fp32 path, only the conversion is gone
  constexpr bool APL_SAMEMAP = ISG_F16X3_APLANES == 3;
#else
  constexpr bool APL = false;
  constexpr bool APL_INTERLEAVED = false;
  constexpr bool APL_SAMEMAP = false;
#endif
  const _Float16 *Ah = reinterpret_cast<const _Float16 *>(A);
  const int64_t a_plane = (int64_t)M * lda;
  hf32x4 ra0v[2], ra1v[2];      // the diagnostic paths stage in a native vector type: a whole-struct copy of HIP's float4 from
                                // a register array into LDS left the four of them in scratch memory (80 bytes per lane)
#define HT_LOAD_TILE(RA, RB, k0)                                                                                 \
  {                                                                                                              \
    if constexpr (APL_SAMEMAP) {   /* the shipped kernel's (row, 16-byte chunk) map on [M][K/32][hi 32 | mid 32] */  \
      /* both iterations written out: a loop over u left RA[u] in SCRATCH memory (80 bytes per lane) */          \
      const int c = tid & 7, gr0 = min(m0 + (tid >> 3), M - 1), gr1 = min(m0 + ((tid + 512) >> 3), M - 1);       \
      RA##v[0] = *reinterpret_cast<const hf32x4 *>(Ah + ((int64_t)gr0 * (lda >> 5) + ((k0) >> 5)) * 64 + c * 8);  \
      RA##v[1] = *reinterpret_cast<const hf32x4 *>(Ah + ((int64_t)gr1 * (lda >> 5) + ((k0) >> 5)) * 64 + c * 8);  \
    } else if constexpr (APL) {                                                                                  \
      const int row = tid >> 2, c8 = tid & 3;                                                                    \
      const int gr = min(m0 + row, M - 1), gk = min((k0) + c8 * 8, K - 8);                                       \
      RA##v[0] = *reinterpret_cast<const hf32x4 *>(                                                              \
          APL_INTERLEAVED ? Ah + ((int64_t)gr * (lda >> 5) + ((k0) >> 5)) * 64 + c8 * 8 : Ah + (int64_t)gr * lda + gk);      \
      RA##v[1] = *reinterpret_cast<const hf32x4 *>(                                                              \
          APL_INTERLEAVED ? Ah + ((int64_t)gr * (lda >> 5) + ((k0) >> 5)) * 64 + 32 + c8 * 8                      \
                          : Ah + a_plane + (int64_t)gr * lda + gk);                                              \
    } else                                                                                                       \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                              \
      const int i = tid + 512 * u;                                                                               \
      const int row = i >> 3, c4 = i & 7;                                                                        \
      const int gr = min(m0 + row, M - 1), gk = min((k0) + c4 * 4, K - 4);                                       \
      RA[u] = *reinterpret_cast<const float4 *>(A + (int64_t)gr * lda + gk);                                     \
    }                                                                                                            \
    {                                                                                                            \
      const int row = tid >> 2, c8 = tid & 3;                                                                    \
      const int gn = min(n0 + row, N - 1);                                                                       \
      _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                              \
        RB[q] = *reinterpret_cast<const hf16x8 *>(Wp + q * plane_stride + (int64_t)gn * ldw + (k0) + c8 * 8);    \
    }                                                                                                            \
  }
  const int nk = Kp / HT_BK;
  const int fr = lane & 31, fk = (lane >> 5) * 8;
#define HT_STEP(RA, RB, kt)                                                                                      \
  {                                                                                                              \
    if ((kt) > 0) __syncthreads();                                                                               \
    if constexpr (APL_SAMEMAP) {                                                                                 \
      const int c = tid & 7;                                                                                     \
      *reinterpret_cast<hf32x4 *>(&sm.a[c >> 2][tid >> 3][(c & 3) * 8]) = RA##v[0];                              \
      *reinterpret_cast<hf32x4 *>(&sm.a[c >> 2][(tid + 512) >> 3][(c & 3) * 8]) = RA##v[1];                      \
    } else if constexpr (APL) {                                                                                  \
      const int row = tid >> 2, c8 = tid & 3;                                                                    \
      *reinterpret_cast<hf32x4 *>(&sm.a[0][row][c8 * 8]) = RA##v[0];                                             \
      *reinterpret_cast<hf32x4 *>(&sm.a[1][row][c8 * 8]) = RA##v[1];                                             \
    } else                                                                                                       \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                              \
      const int i = tid + 512 * u;                                                                               \
      const int row = i >> 3, c4 = i & 7;                                                                        \
      float4 v = RA[u];                                                                                          \
      if (m0 + row >= M || (kt) * HT_BK + c4 * 4 >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);                      \
      v.x *= sa[u]; v.y *= sa[u]; v.z *= sa[u]; v.w *= sa[u];                                                    \
      hf16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};                                  \
      hf16x4 mid = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]), \
                    (_Float16)(v.w - (float)hi[3])};                                                             \
      *reinterpret_cast<hf16x4 *>(&sm.a[0][row][c4 * 4]) = hi;                                                   \
      *reinterpret_cast<hf16x4 *>(&sm.a[1][row][c4 * 4]) = mid;                                                  \
    }                                                                                                            \
    {                                                                                                            \
      const int row = tid >> 2, c8 = tid & 3;                                                                    \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) *reinterpret_cast<hf16x8 *>(&sm.b[q][row][c8 * 8]) = RB[q];  \
    }                                                                                                            \
    __syncthreads();                                                                                             \
    if ((kt) + 2 < nk) HT_LOAD_TILE(RA, RB, ((kt) + 2) * HT_BK)                                                  \
    if constexpr (MF == 16) {                                                                                    \
      hf16x8 a[2][2];                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q)                \
        a[i][q] = *reinterpret_cast<const hf16x8 *>(&sm.a[q][wm * 32 + i * 16 + (lane & 15)][(lane >> 4) * 8]);  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                            \
        hf16x8 b[2];                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                            \
          b[q] = *reinterpret_cast<const hf16x8 *>(&sm.b[q][wn * 64 + j * 16 + (lane & 15)][(lane >> 4) * 8]);   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
          hf32x4 c = acc16[i][j];                                                                                \
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[1], c, 0, 0, 0);                                 \
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][1], b[0], c, 0, 0, 0);                                 \
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][0], b[0], c, 0, 0, 0);                                 \
          acc16[i][j] = c;                                                                                       \
        }                                                                                                        \
      }                                                                                                          \
    } else                                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
      hf16x8 a[2], b[2][2];                                                                                      \
      _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                              \
        a[q] = *reinterpret_cast<const hf16x8 *>(&sm.a[q][wm * 32 + fr][ks * 16 + fk]);                          \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int q = 0; q < 2; ++q)                \
        b[j][q] = *reinterpret_cast<const hf16x8 *>(&sm.b[q][wn * 64 + j * 32 + fr][ks * 16 + fk]);              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                            \
        hf32x16 c = acc[j];                                                                                      \
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][1], c, 0, 0, 0);                                   \
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[j][0], c, 0, 0, 0);                                   \
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[j][0], c, 0, 0, 0);                                   \
        acc[j] = c;                                                                                              \
      }                                                                                                          \
    }                                                                                                            \
  }
  HT_LOAD_TILE(ra0, rb0, 0)
  if (nk > 1) HT_LOAD_TILE(ra1, rb1, HT_BK)
  for (int kt = 0; kt < nk; kt += 2) {
    HT_STEP(ra0, rb0, kt)
    if (kt + 1 < nk) HT_STEP(ra1, rb1, kt + 1)
  }
#undef HT_STEP
#undef HT_LOAD_TILE

  // ---- epilogue: scale back, + bias, activation, per-wave LDS patch -> 16-byte row stores (+ the row maxima of D) ------
  float inv_a[16];
  if constexpr (MF == 16) {      // [i * 4 + r]: row i * 16 + 4 * (lane >> 4) + r of the wave's 32
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) inv_a[i * 4 + r] = sm.inv[wm * 32 + i * 16 + 4 * (lane >> 4) + r];
  } else {
    const int h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) inv_a[r] = sm.inv[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
  }
  __syncthreads();   // every wave is done with the operand tiles and the scale table: the LDS is free
  float *patch = reinterpret_cast<float *>(&sm) + wave * (32 * 36);
  const int h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int colb = n0 + wn * 64 + j * 32;
    if constexpr (MF == 16) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int cin = jj * 16 + (lane & 15);
        const float wi = w_inv[min(colb + cin, N - 1)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            patch[(i * 16 + 4 * (lane >> 4) + r) * 36 + cin] = (acc16[i][2 * j + jj][r] * inv_a[i * 4 + r]) * wi;
      }
    } else {
      const int col = colb + fr;
      const float wi = w_inv[min(col, N - 1)];
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + fr] = (acc[j][r] * inv_a[r]) * wi;
    }
    __builtin_amdgcn_wave_barrier();
    const int rowb = m0 + wm * 32;
    const bool vec_ok = (ldd & 3) == 0 && colb + 32 <= N && (reinterpret_cast<uintptr_t>(D) & 15) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = q * 64 + lane;
      const int rr = idx >> 3, c4 = idx & 7;
      float4 v = *reinterpret_cast<const float4 *>(&patch[rr * 36 + c4 * 4]);
      const int row = rowb + rr;
      const int cc = colb + c4 * 4;
      if (ACCUM && row < M) {
        const float *src = D + (int64_t)row * ldd + cc;
        if (vec_ok) {
          const float4 o = *reinterpret_cast<const float4 *>(src);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        } else {
          if (cc + 0 < N) v.x += src[0];
          if (cc + 1 < N) v.y += src[1];
          if (cc + 2 < N) v.z += src[2];
          if (cc + 3 < N) v.w += src[3];
        }
      }
      if (bias) {
        v.x += cc + 0 < N ? bias[cc + 0] : 0.f;
        v.y += cc + 1 < N ? bias[cc + 1] : 0.f;
        v.z += cc + 2 < N ? bias[cc + 2] : 0.f;
        v.w += cc + 3 < N ? bias[cc + 3] : 0.f;
      }
      if (ACT == 1) {
        const isg_f32x2 g0 = gelu_exact2(isg_f32x2{v.x, v.y}), g1 = gelu_exact2(isg_f32x2{v.z, v.w});
        v.x = g0.x; v.y = g0.y; v.z = g1.x; v.w = g1.y;
      }
      if (ACT == 2) {     // ReLU (a NaN stays a NaN, as in torch)
        v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
      }
      if (RMAX) {   // largest magnitude of this row's (up to) 32 columns; columns beyond N are not part of D
        const int c0 = colb + c4 * 4;
        float mx = c0 < N ? fabsf(v.x) : 0.f;
        if (c0 + 1 < N) mx = fmaxf(mx, fabsf(v.y));
        if (c0 + 2 < N) mx = fmaxf(mx, fabsf(v.z));
        if (c0 + 3 < N) mx = fmaxf(mx, fabsf(v.w));
        mx = fmaxf(mx, dpp_mov<ISG_DPP_XOR1>(mx));
        mx = fmaxf(mx, dpp_mov<ISG_DPP_XOR2>(mx));
        mx = fmaxf(mx, dpp_mov<ISG_DPP_HMIRROR>(mx));
        if (c4 == 0 && row < M && colb < N) d_rowmax[(int64_t)row * ((N + 31) / 32) + (colb >> 5)] = mx;
      }
      if (row < M) {
        float *dst = D + (int64_t)row * ldd + colb + c4 * 4;
        if (vec_ok) {
          if (nt_store) {
            typedef float ht_f32x4 __attribute__((ext_vector_type(4)));
            ht_f32x4 w4 = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w4, reinterpret_cast<ht_f32x4 *>(dst));
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
}

}  // namespace isg

extern "C" int isg_split_f16x2_rows(const float *w, int64_t rows, int32_t K, uint16_t *planes, float *inv_scale,
                                    void *stream) {
  if (rows < 0 || K <= 0) return ISG_EINVAL;
  if (rows == 0) return ISG_OK;
  if (!w || !planes || !inv_scale) return ISG_EINVAL;
  if (rows >= (1ll << 24)) return ISG_EUNSUPPORTED;
  const int Kp = (K + HT_BK - 1) / HT_BK * HT_BK;
  split_f16x2_rows_kernel<<<(unsigned)rows, 64, 0, as_stream(stream)>>>(w, (int)rows, K, Kp,
                                                                        reinterpret_cast<_Float16 *>(planes), inv_scale);
  return check_launch();
}

// a_rowmax fp32 [M, P]: partial maxima of |a| per row (P >= 1); d_rowmax NULL or fp32 [M, ceil(N / 32)] (written).
// k_offset / accumulate: this call covers columns [k_offset, k_offset + K) of a weight whose planes have `K_total` columns
// (k_offset a multiple of 32); with accumulate != 0 the result is added to what `d` holds before bias and activation.
extern "C" int isg_linear_f16x3_tile(const float *a, const float *a_rowmax, int32_t P, int32_t ldp,
                                     const uint16_t *w_planes, const float *w_inv_scale, const float *bias, float *d,
                                     float *d_rowmax, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                                     int32_t K_total, int32_t k_offset, int32_t accumulate, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || lda < K || ldd < N || act < 0 || act > 2 || P <= 0 || ldp < P) return ISG_EINVAL;
  if (K_total < K || k_offset < 0 || k_offset + K > K_total || (k_offset & 31)) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a || !a_rowmax || !w_planes || !w_inv_scale || !d) return ISG_EINVAL;
  if ((K & 3) != 0 || (lda & 3) != 0 || (reinterpret_cast<uintptr_t>(a) & 15) != 0 || M >= (1ll << 31) || K > 1024 || P > 64)
    return ISG_EUNSUPPORTED;
  const int Kp = (K + HT_BK - 1) / HT_BK * HT_BK;
  const int ldw = (K_total + HT_BK - 1) / HT_BK * HT_BK;
  const long long mt = (M + HT_BM - 1) / HT_BM;
  if (mt > 65535) return ISG_EUNSUPPORTED;
  dim3 grid((unsigned)((N + HT_BN - 1) / HT_BN), (unsigned)mt), block(512);
  static const long long nt_mb = [] { const char *e = getenv("ISG_GEMM_NT_MB"); return e ? atoll(e) : 128ll; }();
  const int nt = nt_mb >= 0 && (long long)M * N * 4 >= nt_mb * 1000000ll;
  dim3 gridx(grid.x, (grid.y + 7) / 8 * 8);
  const bool xcd = grid.x > 1;
  const _Float16 *wp = reinterpret_cast<const _Float16 *>(w_planes) + k_offset;
  hipStream_t st = as_stream(stream);
  static const int mf = [] { const char *e = getenv("ISG_F16X3_MFMA"); return e && atoi(e) == 32 ? 32 : 16; }();
#define ISG_HT4(ACT_, X_, R_, C_, MF_) linear_f16x3_tile_kernel<ACT_, X_, R_, C_, MF_><<<(X_) ? gridx : grid, block, 0, st>>>(a, a_rowmax, P, ldp, wp, w_inv_scale, bias, d, d_rowmax, (int)M, N, K, Kp, ldw, lda, ldd, nt)
#define ISG_HT3(ACT_, R_, C_)                                                                                     \
  do {                                                                                                            \
    if (xcd) { if (mf == 16) ISG_HT4(ACT_, true, R_, C_, 16); else ISG_HT4(ACT_, true, R_, C_, 32); }              \
    else { if (mf == 16) ISG_HT4(ACT_, false, R_, C_, 16); else ISG_HT4(ACT_, false, R_, C_, 32); }                \
  } while (0)
#define ISG_HT(ACT_, R_) do { if (accumulate) ISG_HT3(ACT_, R_, true); else ISG_HT3(ACT_, R_, false); } while (0)
  if (act == 1) { if (d_rowmax) ISG_HT(1, true); else ISG_HT(1, false); }
  else if (act == 2) { if (d_rowmax) ISG_HT(2, true); else ISG_HT(2, false); }
  else { if (d_rowmax) ISG_HT(0, true); else ISG_HT(0, false); }
#undef ISG_HT
#undef ISG_HT3
#undef ISG_HT4
  return check_launch();
}

// rowmax[m] = max_k |a[m, k]|: the row scales of isg_linear_f16x3_tile for an input whose producer did not leave them.
// One extra pass over `a` (M*K*4 bytes): worth it when the Linear is wide (the caller's policy: N >= 512).
namespace isg {
__global__ __launch_bounds__(256) void row_absmax_kernel(const float *__restrict__ a, int M, int K, int lda,
                                                         float *__restrict__ rowmax) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float4 *r4 = reinterpret_cast<const float4 *>(a + (int64_t)row * lda);
  float mx = 0.f;
  for (int c = lane; c < (K >> 2); c += 64) {
    const float4 v = r4[c];
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  mx = wave_max(mx);
  if (lane == 0) rowmax[row] = mx;
}
}  // namespace isg

extern "C" int isg_row_absmax(const float *a, int64_t M, int32_t K, int32_t lda, float *rowmax, void *stream) {
  if (M < 0 || K <= 0 || lda < K) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a || !rowmax) return ISG_EINVAL;
  if ((K & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(a) & 15) || (M + 3) / 4 >= (1ll << 31)) return ISG_EUNSUPPORTED;
  row_absmax_kernel<<<(unsigned)((M + 3) / 4), 256, 0, as_stream(stream)>>>(a, (int)M, K, lda, rowmax);
  return check_launch();
}
