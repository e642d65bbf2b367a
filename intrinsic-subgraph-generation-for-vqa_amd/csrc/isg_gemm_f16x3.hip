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
#include "isg_common.hpp"

#include <stdlib.h>

namespace isg {

typedef __attribute__((ext_vector_type(8))) _Float16 hf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 hf16x4;
typedef __attribute__((ext_vector_type(16))) float hf32x16;

constexpr int H3_BM = 64, H3_KC = 128, H3_LD = H3_KC + 8, H3_THREADS = 256;

// power of two that moves |mx| into [2^13, 2^14), and its inverse; 1 for zero / non-finite rows
__device__ __forceinline__ void h3_scale(float mx, float &s, float &inv) {
  const int e = (int)((__float_as_uint(mx) >> 23) & 255u);      // biased exponent
  if (e == 0 || e == 255) { s = 1.f; inv = 1.f; return; }
  s = __uint_as_float((unsigned)(127 + 13 + 127 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 13) << 23);
}

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

template <int ACT, int WTN>
__global__ __launch_bounds__(H3_THREADS, 2) void linear_f16x3_kernel(const float *__restrict__ A,
                                                                     const _Float16 *__restrict__ Wf,
                                                                     const float *__restrict__ w_inv,
                                                                     const float *__restrict__ bias, float *__restrict__ D,
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
      float *dst = D + obase + (int64_t)row * ldd + col;                                                         \
      if (nt_store) __builtin_nontemporal_store(v, dst);                                                         \
      else *dst = v;                                                                                             \
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

extern "C" int isg_linear_f16x3(const float *a, const uint16_t *w_frag, const float *w_inv_scale, const float *bias,
                                float *d, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
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
  const int nt = nt_b >= 0 && (long long)M * N * 4 >= nt_b;
  const _Float16 *wf = reinterpret_cast<const _Float16 *>(w_frag);
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)panels), block(H3_THREADS);
#define ISG_H3(ACT_, W_) linear_f16x3_kernel<ACT_, W_><<<grid, block, 0, st>>>(a, wf, w_inv_scale, bias, d, (int)M, N, K, KS, NT, lda, ldd, nt, out_cols, (long long)out_stride)
  if (act == 1) { if (wtn == 2) ISG_H3(1, 2); else ISG_H3(1, 1); }
  else { if (wtn == 2) ISG_H3(0, 2); else ISG_H3(0, 1); }
#undef ISG_H3
  return check_launch();
}
