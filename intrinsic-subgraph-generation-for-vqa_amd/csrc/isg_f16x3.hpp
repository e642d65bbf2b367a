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

// ---- planes32 (csrc/isg_gemm_h3p.hip): row r, k-tile kt = one 128-byte line [hi 32 | mid 32] fp16 of the row times its scale ----
// One row -> its planes32 lines, by a group of LPR lanes (wave per row for wide rows; sixteen lanes per row up to 128 float4 --
// K = 300 is 75 float4: five passes of 16 lanes keep 94 % of the lane slots busy, a whole wave 59 %).  `val(c)` = float4 c of the
// row; with PMAX > 0 the row is held in registers between its largest magnitude and its split, else it is evaluated twice.
template <int LPR, int PMAX, typename F, typename G>
__device__ __forceinline__ void planes32_row(int l, int nc, int KT, _Float16 *__restrict__ p, float *__restrict__ inv_slot, F val, G keep) {
  float4 v[PMAX > 0 ? PMAX : 1];
  float mx = 0.f;
  if constexpr (PMAX > 0) {
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
      const int c = l + LPR * q;
      v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < nc) v[q] = val(c);
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w))));
    }
  } else {
    for (int c = l; c < nc; c += LPR) {
      const float4 t = val(c);
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))));
    }
  }
  mx = group_max<LPR>(mx);
  float s, inv;
  h3_scale(mx, s, inv);
  if (l == 0) *inv_slot = inv;
  auto put = [&](int c, float4 t) {
    if (c < nc) keep(c, t);
    t.x *= s; t.y *= s; t.z *= s; t.w *= s;
    const hf16x4 hi = {(_Float16)t.x, (_Float16)t.y, (_Float16)t.z, (_Float16)t.w};
    const hf16x4 mid = {(_Float16)(t.x - (float)hi[0]), (_Float16)(t.y - (float)hi[1]), (_Float16)(t.z - (float)hi[2]),
                        (_Float16)(t.w - (float)hi[3])};
    _Float16 *d = p + (c >> 3) * 64 + (c & 7) * 4;
    *reinterpret_cast<hf16x4 *>(d) = hi;
    *reinterpret_cast<hf16x4 *>(d + 32) = mid;
  };
  if constexpr (PMAX > 0) {
#pragma unroll
    for (int q = 0; q < PMAX; ++q)
      if (l + LPR * q < KT * 8) put(l + LPR * q, v[q]);         // beyond nc: the zeros of the k padding
  } else {
    for (int c = l; c < KT * 8; c += LPR) put(c, c < nc ? val(c) : make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

}  // namespace isg
