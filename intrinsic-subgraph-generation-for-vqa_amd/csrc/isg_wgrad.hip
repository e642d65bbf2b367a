// Weight gradient of a linear layer: dW[N,K] = g^T x, g fp32 [M,N] (gradient of the output), x fp32 [M,K] (the input).
//
// The contraction runs over the ROWS (M ~ 1e5 nodes / edges) and the result is tiny, so this is a split-M GEMM:
// grid = (N/128) x (K/128) x S, every workgroup reduces its slice of rows into a 128 x 128 partial tile, partials are
// summed afterwards (fixed order, no atomics).  It uses the fp32-input matrix core instruction
// v_mfma_f32_32x32x2_f32 on purpose: its operand layout (lane -> A[i = lane & 31][k = lane >> 5]) wants, for both
// operands, one fp32 per lane taken along a ROW of g / x -- exactly how the row-major activations sit in memory and in
// LDS -- whereas the bf16 instruction wants 8 consecutive values along the contraction axis, i.e. a transpose of both
// operands.  Products are exact fp32 x fp32 with fp32 accumulation.  (The fp32 MFMA peak is 157 TF; the fp32 hipBLASLt
// kernels torch picks for these shapes reach ~50.)
#include "isg_common.hpp"

#include <stdlib.h>

namespace isg {

typedef __attribute__((ext_vector_type(16))) float wg_f32x16;

constexpr int WG_T = 128;        // output tile (n and k)
constexpr int WG_BM = 32;        // rows per step
constexpr int WG_LD = WG_T + 4;  // LDS row pitch in floats

__global__ __launch_bounds__(512, 2) void wgrad_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                       float *__restrict__ partial, int M, int N, int K, int ldg,
                                                       int ldx, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) float sG[2][WG_BM][WG_LD];
  __shared__ __attribute__((aligned(16))) float sX[2][WG_BM][WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int n0 = blockIdx.x * WG_T, k0 = blockIdx.y * WG_T;
  const int m_begin = blockIdx.z * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  wg_f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

  // a tile = 32 rows x 128 cols = 1024 float4: 2 per thread (rows r and r + 16, float4 column c4)
  const int lr = tid >> 5, c4 = tid & 31;
  const bool g_vec = (ldg & 3) == 0 && n0 + WG_T <= N && (reinterpret_cast<uintptr_t>(g) & 15) == 0;
  const bool x_vec = (ldx & 3) == 0 && k0 + WG_T <= K && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  auto load4 = [&](const float *base, int ld, int row, int col0, int ncols, bool vec) -> float4 {
    if (row >= m_end) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float *p = base + (size_t)row * ld + col0;
    if (vec) return *reinterpret_cast<const float4 *>(p);
    float4 v;
    v.x = col0 + 0 < ncols ? p[0] : 0.f;
    v.y = col0 + 1 < ncols ? p[1] : 0.f;
    v.z = col0 + 2 < ncols ? p[2] : 0.f;
    v.w = col0 + 3 < ncols ? p[3] : 0.f;
    return v;
  };
  float4 rg[2], rx[2];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rg[u] = load4(g, ldg, m0 + lr + 16 * u, n0 + c4 * 4, N, g_vec);
      rx[u] = load4(x, ldx, m0 + lr + 16 * u, k0 + c4 * 4, K, x_vec);
    }
  };
  const int steps = (m_end - m_begin + WG_BM - 1) / WG_BM;
  if (steps > 0) fetch(m_begin);
  const int ar = lane >> 5, ac = lane & 31;
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<float4 *>(&sG[buf][lr + 16 * u][c4 * 4]) = rg[u];
      *reinterpret_cast<float4 *>(&sX[buf][lr + 16 * u][c4 * 4]) = rx[u];
    }
    __syncthreads();                       // tile s visible; tile s-1 (other buffer) fully consumed before it is rewritten
    if (s + 1 < steps) fetch(m_begin + (s + 1) * WG_BM);
#pragma unroll
    for (int mm = 0; mm < WG_BM / 2; ++mm) {
      const float a = sG[buf][2 * mm + ar][wm * 32 + ac];
      const float b0 = sX[buf][2 * mm + ar][wn * 64 + ac];
      const float b1 = sX[buf][2 * mm + ar][wn * 64 + 32 + ac];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
  }
  // partial[z][n][k]; acc element r of a lane: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
  float *out = partial + (size_t)blockIdx.z * N * K;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * ar;
    if (n < N) {
      const int ka = k0 + wn * 64 + ac, kb = ka + 32;
      if (ka < K) out[(size_t)n * K + ka] = acc0[r];
      if (kb < K) out[(size_t)n * K + kb] = acc1[r];
    }
  }
}

}  // namespace isg

using namespace isg;

extern "C" int64_t isg_linear_wgrad_splits(int64_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int64_t tiles = (int64_t)((N + WG_T - 1) / WG_T) * ((K + WG_T - 1) / WG_T);
  const char *env = getenv("ISG_WGRAD_WGS");              // tuning switch: total workgroups aimed at
  const int64_t target = env ? atoll(env) : 512;           // one resident round: 256 CUs x 2 workgroups (measured best)
  int64_t s = (target + tiles - 1) / tiles;
  const int64_t by_rows = (M + 255) / 256;                // at least 256 rows per split
  if (s > by_rows) s = by_rows;
  if (s < 1) s = 1;
  if (s > 65535) s = 65535;
  return s;
}

extern "C" int isg_linear_wgrad(const float *grad_out, const float *x, float *partial, int64_t M, int32_t N, int32_t K,
                                int32_t ldg, int32_t ldx, int64_t splits, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || ldg < N || ldx < K || splits <= 0) return ISG_EINVAL;
  if (!grad_out || !x || !partial) return ISG_EINVAL;
  if (M >= (1ll << 31) || splits > 65535 || (N + WG_T - 1) / WG_T > 65535) return ISG_EUNSUPPORTED;
  int64_t rows = (M + splits - 1) / splits;
  rows = (rows + WG_BM - 1) / WG_BM * WG_BM;
  dim3 grid((unsigned)((N + WG_T - 1) / WG_T), (unsigned)((K + WG_T - 1) / WG_T), (unsigned)splits), block(512);
  wgrad_kernel<<<grid, block, 0, as_stream(stream)>>>(grad_out, x, partial, (int)M, N, K, ldg, ldx, (int)rows);
  return check_launch();
}
