// Small-M fp32 Linear: D[M,N] = act(A[M,K] . W[N,K]^T + bias) for the LATENCY-bound regime.
//
// Reference call sites: every torch.nn.Linear / PyG Linear of the path (ISubGVQA/models/question_encoder.py:20-25, question_decoder.py:
// 25-30, scene_graph_encoder.py:108-143, mgat_v2_conv.py:177-181, mgat.py:156, masking.py:137,152, att_pooling.py:62,66,
// isubgvqa.py:247,265,288-292) when the batch is a handful of questions -- run_token_coo.py:49-79 evaluates ONE question per forward.
//
// Why a kernel of its own.  The exact-split tile kernels (isg_gemm.hip, isg_gemm_f16x3.hip, isg_gemm_h3p.hip) are built for
// throughput: a workgroup walks the whole of K in 32-wide steps, every step a global load -> LDS -> barrier -> MFMA round trip.  At
// M = 96 rows (8 questions of 12 tokens) a 512 x 512 Linear is one such chain of 16 steps on a few workgroups: 20 us whatever the
// size, a K = 2048 reduction four launches of it (84 us), and the full model is ~75 of them back to back: 2.8 ms for ONE question
// as for 256 (profiles/r06_k_full_model_b8_kernel_stats.csv).  Here the reduction is split over the EIGHT WAVES of a workgroup
// (wave w owns k in [w ks, (w + 1) ks)), every wave reads its slices of A and W straight from memory into MFMA operands -- no LDS
// staging, no barrier inside the k loop -- and the eight partial tiles are added through LDS in wave order: one barrier per
// workgroup.  The products are TRUE fp32 (v_mfma_f32_16x16x4_f32: 1/16 of the fp16 rate, irrelevant at these sizes): no row
// scales, no planes, no weight preparation.
//   grid    (ceil(N / 16), ceil(M / 16)); block 512 = 8 waves; LDS 8 x 16 x 17 x 4 = 8,704 bytes.  16 x 16 output tiles, not 32 x 32:
//           the fp32 MFMA rate of ONE CU is what a workgroup's chain costs (a 32 x 32 x 2048 tile is 8 us of it), and a small Linear has
//           few tiles -- four times as many workgroups put four times as many CUs under the same reduction
//   wave    lane (r = lane & 15, q = lane >> 4) loads A[m0 + r][k .. k + 3] and W[n0 + r][k .. k + 3] with k = kk + 4 q as float4 and
//           issues four 16x16x4 MFMAs on their components: MFMA j of a step reduces k = kk + 4 q + j over q = 0..3
//   order   a row's sum is: per wave its k slice ascending (as above), then waves 0..7 ascending -- a function of K alone, so the
//           SAME row gives the SAME bits wherever it sits in the batch and whatever the batch's size (distributed.py's contract)
#include "isg_common.hpp"

namespace isg {

constexpr int SK_WAVES = 8, SK_THREADS = 64 * SK_WAVES, SK_T = 16, SK_LD = SK_T + 1;
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

template <int ACT>
__global__ __launch_bounds__(SK_THREADS) void linear_skinny_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                                   const float *__restrict__ bias, float *__restrict__ D, int M, int N,
                                                                   int K, int lda, int ldw, int ldd) {
  __shared__ float s_part[SK_WAVES][SK_T][SK_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * SK_T, m0 = blockIdx.y * SK_T;
  // rows past the end repeat the last row (their results are never stored)
  const float *ap = A + (int64_t)min(m0 + r, M - 1) * lda;
  const float *wp = W + (int64_t)min(n0 + r, N - 1) * ldw;
  const int ks = (K + 16 * SK_WAVES - 1) / (16 * SK_WAVES) * 16;         // a wave's slice: a whole number of 16-wide steps
  const int k_lo = wave * ks, k_hi = min(K, k_lo + ks);
  sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const sk_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // four steps (64 columns of k) per round; the NEXT round's eight loads are in flight under this round's sixteen MFMAs (two
  // register images, the loop unrolled by two so that neither is ever copied)
  sk_f32x4 a0[4], b0[4], a1[4], b1[4];
#define SK_LOAD(RA, RB, kk_)                                                                                       \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                  \
    const int k = (kk_) + 16 * u + 4 * q;                               /* 4 | K: k < K means k + 3 < K */          \
    const bool in = k < k_hi;                                                                                      \
    RA[u] = in ? *reinterpret_cast<const sk_f32x4 *>(ap + k) : zero;                                               \
    RB[u] = in ? *reinterpret_cast<const sk_f32x4 *>(wp + k) : zero;                                               \
  }
#define SK_MFMA(RA, RB)                                                                                            \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(RA[u][j], RB[u][j], acc, 0, 0, 0); \
  }
  if (k_lo < k_hi) {
    SK_LOAD(a0, b0, k_lo)
#pragma unroll 1
    for (int kk = k_lo; kk < k_hi; kk += 128) {
      SK_LOAD(a1, b1, kk + 64)            // (past the slice's end: zeros, and the MFMAs on them below would add nothing)
      SK_MFMA(a0, b0)
      if (kk + 64 >= k_hi) break;
      SK_LOAD(a0, b0, kk + 128)
      SK_MFMA(a1, b1)
    }
  }
#undef SK_LOAD
#undef SK_MFMA
  // acc[i]: row 4 q + i of the tile, column r
#pragma unroll
  for (int i = 0; i < 4; ++i) s_part[wave][4 * q + i][r] = acc[i];
  __syncthreads();
  if (tid < SK_T * SK_T) {
    const int row = tid >> 4, col = tid & 15;
    float v = s_part[0][row][col];
#pragma unroll
    for (int w = 1; w < SK_WAVES; ++w) v += s_part[w][row][col];
    const int m = m0 + row, n = n0 + col;
    if (m < M && n < N) {
      if (bias) v += bias[n];
      if (ACT == 1) v = gelu_exact(v);
      if (ACT == 2) v = fmaxf(v, 0.f);
      D[(int64_t)m * ldd + n] = v;
    }
  }
}

}  // namespace isg

using namespace isg;

// d[M,N] = act(a[M,K] @ w[N,K]^T + bias), all fp32, w in torch's Linear layout (row stride ldw): see the file header.
// act 0 none, 1 exact GELU, 2 ReLU.  ISG_EUNSUPPORTED unless 4 | K, 4 | lda, 4 | ldw, a and w 16-byte aligned, M <= 65535 x 32 rows (grid.y).
extern "C" int isg_linear_skinny(const float *a, int32_t lda, const float *w, int32_t ldw, const float *bias, float *d, int32_t ldd,
                                 int64_t M, int32_t N, int32_t K, int32_t act, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldd < N || act < 0 || act > 2) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a || !w || !d) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if ((K & 3) || (lda & 3) || (ldw & 3) || mis(a) || mis(w) || M > 65535ll * SK_T) return ISG_EUNSUPPORTED;
  const dim3 grid((unsigned)((N + SK_T - 1) / SK_T), (unsigned)((M + SK_T - 1) / SK_T));
  hipStream_t st = as_stream(stream);
  if (act == 1) linear_skinny_kernel<1><<<grid, SK_THREADS, 0, st>>>(a, w, bias, d, (int)M, N, K, lda, ldw, ldd);
  else if (act == 2) linear_skinny_kernel<2><<<grid, SK_THREADS, 0, st>>>(a, w, bias, d, (int)M, N, K, lda, ldw, ldd);
  else linear_skinny_kernel<0><<<grid, SK_THREADS, 0, st>>>(a, w, bias, d, (int)M, N, K, lda, ldw, ldd);
  return check_launch();
}
