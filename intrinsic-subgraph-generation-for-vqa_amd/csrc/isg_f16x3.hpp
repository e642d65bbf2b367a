// Shared pieces of the fp16 three-product form (isg_gemm_f16x3.hip, isg_mp_logits.hip): vector types and the per-row
// power-of-two scale that moves a row's largest magnitude into [2^13, 2^14).
#pragma once
#include "isg_common.hpp"

namespace isg {

typedef __attribute__((ext_vector_type(8))) _Float16 hf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 hf16x4;
typedef __attribute__((ext_vector_type(16))) float hf32x16;
typedef __attribute__((ext_vector_type(4))) float hf32x4;

// power of two that moves |mx| into [2^13, 2^14), and its inverse; 1 for zero / subnormal / non-finite rows.
// Rows below 2^-113 (biased exponent < 14) are left unscaled as well: the scale 2^(140 - e) would not fit an fp32 exponent
// (e = 12 gave s = +inf, inv = -inf and a NaN row; e < 12 wrapped to -0.0), and at that magnitude the fp16 planes are all
// zero either way -- the row contributes ~1e-34 x |w|, far below the accumulation's own rounding.
__device__ __forceinline__ void h3_scale(float mx, float &s, float &inv) {
  const int e = (int)((__float_as_uint(mx) >> 23) & 255u);      // biased exponent
  if (e < 14 || e == 255) { s = 1.f; inv = 1.f; return; }
  s = __uint_as_float((unsigned)(127 + 13 + 127 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 13) << 23);
}

}  // namespace isg
