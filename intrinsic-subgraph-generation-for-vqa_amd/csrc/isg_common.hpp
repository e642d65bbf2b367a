// Shared device helpers for libisg_hip.so (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "../../include/isg.h"

#define ISG_WAVE 64

namespace isg {

// ---- host-side launch bookkeeping -------------------------------------------------------------
int check_launch();                 // hipGetLastError() -> ISG_OK / ISG_ELAUNCH (records the text)
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
// Per-DEVICE state of this process.  More than 64 KB of dynamic LDS is an attribute of (function, device) and a persistent grid
// is sized from the device's CU count: a process that drives several GPUs must not reuse the first device's answers.
inline int current_device() {
  int d = 0;
  return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : -1;
}
inline int device_cus() {
  static int cus[64] = {};
  const int d = current_device();
  if (d < 0) return 256;
  if (!cus[d]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    cus[d] = n;
  }
  return cus[d];
}
template <auto Kernel>
inline bool dyn_lds_ok(int bytes) {       // hipFuncAttributeMaxDynamicSharedMemorySize per (kernel, device): raised whenever a
  static int granted[64] = {};            // caller asks for more than the largest size set so far (0 nothing set yet, -1 refused) --
  const int d = current_device();         // a first caller with a small shape must not cap a later, larger one
  if (d < 0 || bytes < 0) return false;
  if (granted[d] < 0) return false;
  if (bytes > granted[d] || granted[d] == 0) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
      if (granted[d] == 0) granted[d] = -1;        // never worked on this device: do not ask again on every launch
      return false;
    }
    granted[d] = bytes > 0 ? bytes : 1;
  }
  return true;
}

// ---- cross-lane movement via DPP (no LDS traffic) --------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
// DPP controls: quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140
#define ISG_DPP_XOR1 0xB1
#define ISG_DPP_XOR2 0x4E
#define ISG_DPP_HMIRROR 0x141
#define ISG_DPP_MIRROR 0x140

// Sum over aligned groups of G lanes (G in {8,16,32,64}); every lane of a group gets the group total.
// The butterfly order is fixed, so the result is bitwise reproducible run to run.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<ISG_DPP_XOR1>(v);
  v += dpp_mov<ISG_DPP_XOR2>(v);
  v += dpp_mov<ISG_DPP_HMIRROR>(v);
  if (G >= 16) v += dpp_mov<ISG_DPP_MIRROR>(v);
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, dpp_mov<ISG_DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<ISG_DPP_XOR2>(v));
  v = fmaxf(v, dpp_mov<ISG_DPP_HMIRROR>(v));
  if (G >= 16) v = fmaxf(v, dpp_mov<ISG_DPP_MIRROR>(v));
  if (G >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  if (G >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}

// One rounding per operation, whatever the compiler would like to contract: the reference's CPU kernels add and multiply in
// separate steps (scatter sums, GraphNorm's statistics, the weighted aggregation), and results that must match them -- or each
// other, between a per-graph kernel and its tile form -- cannot depend on where the backend finds an fma.  hipcc compiles device
// code with -ffp-contract=fast-honor-pragmas, and its __fmul_rn / __fadd_rn / __fsub_rn are plain `x * y`, `x + y`, `x - y`
// (__clang_hip_math.h) that the backend fuses at will -- the same source line came out as v_pk_mul + v_add in one kernel and as
// v_pk_fma in its neighbour.  The pragma takes the `contract` flag off these operations, so nothing fuses with them.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ double dmul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double dadd_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ double dsub_rn(double a, double b) {
#pragma clang fp contract(off)
  return a - b;
}

// A value in a 32-bit register of its own.  Packed fp32 operations must never take a LOW-half operand from the HIGH dword of a
// register pair on this chip (v_pk_*_f32 with op_sel:[..1..]: DESIGN.md 16.1 -- intermittently loses the low-half result of 16
// lanes; tools/scan_pk_cross.py fails the CPU suite on one); the compiler emits that form when a packed operation consumes one
// element of a value it holds as a pair.  own_reg() between the pair and its scalar use leaves it nothing to select from.
__device__ __forceinline__ float own_reg(float x) {
  asm("" : "+v"(x));
  return x;
}

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// The same with four products and three sums each rounded, left to right: written out so that the bits do not depend on which
// products the compiler happens to contract or vectorize in a given loop (one source line gave unfused packed products in one
// kernel and v_pk_fma chains in its neighbour).  For the per-graph attention logits and gates, whose per-graph and tile kernels
// must agree bit for bit (isg_norm_pool.hip / isg_layer_tile.hip / isg_layer_conv.hip / isg_sampler.hip).
__device__ __forceinline__ float dot4_rn(const float4 &a, const float4 &b) {
  return add_rn(add_rn(add_rn(mul_rn(a.x, b.x), mul_rn(a.y, b.y)), mul_rn(a.z, b.z)), mul_rn(a.w, b.w));
}

// exact (erf) GELU, the torch.nn.functional.gelu default used everywhere in the reference: 0.5 x (1 + erf(x / sqrt 2)).
// GELU epilogues are VALU-bound (profiles/r03_c_dense_tail_stamps.txt: libm's erff is ~60 instructions with both of its
// branches live in a wave), so the factor 1 + erf is formed without erf:  with t = |x| / sqrt 2 and e = erfc(t),
//     1 + erf(x / sqrt 2) = 2 - e  (x > 0),   = e  (x <= 0)                     [no cancellation on the negative side]
// and e = exp2(t R(t)) with ONE degree-8 polynomial R on [0, 6] (weighted minimax fit of log2 erfc, |e - erfc| < 3e-9 in exact
// arithmetic) and the hardware exponential v_exp_f32.  Absolute accuracy is what the 1 + erf form can use: measured in fp32
// against a float64 GELU over [-12, 12], |error| / max(|x|, 1) <= 1.2e-7 (the libm formula: 1.1e-7), <= 4 ulp for x > 0, and
// 5x closer than the libm formula for x < 0 (which loses erf's bits in 1 + erf).  Beyond t = 6 (|x| > 8.5) e < 2e-17.
// Written on float pairs so that the polynomial runs on v_pk_fma_f32.  -DISG_GELU_LIBM restores libm's erff (A/B).
typedef float isg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ isg_f32x2 gelu_exact2(isg_f32x2 x) {
#ifdef ISG_GELU_LIBM
  return isg_f32x2{0.5f * x.x * (1.0f + erff(x.x * 0.70710678118654752440f)),
                   0.5f * x.y * (1.0f + erff(x.y * 0.70710678118654752440f))};
#else
  isg_f32x2 t = isg_f32x2{fabsf(x.x), fabsf(x.y)} * 0.70710678118654752440f;
  t = isg_f32x2{fminf(t.x, 6.0f), fminf(t.y, 6.0f)};
  isg_f32x2 r = isg_f32x2{1.1604608516790904e-05f, 1.1604608516790904e-05f};
  r = r * t + (-0.00015296229685191065f);
  r = r * t + 0.0008482258417643607f;
  r = r * t + (-0.002274767030030489f);
  r = r * t + 8.478287054458633e-05f;
  r = r * t + 0.02772449143230915f;
  r = r * t + (-0.1483079195022583f);
  r = r * t + (-0.9184429049491882f);
  r = r * t + (-1.6279072761535645f);
  const isg_f32x2 q = r * t;
  const float e0 = __builtin_amdgcn_exp2f(q.x), e1 = __builtin_amdgcn_exp2f(q.y);
  const isg_f32x2 fac = isg_f32x2{x.x > 0.f ? 2.0f - e0 : e0, x.y > 0.f ? 2.0f - e1 : e1};
  return (x * 0.5f) * fac;
#endif
}
// libm form: the node gate's scores decide the top-k masks (bit-exact vs the CPU path), one value per node: not worth a ulp
__device__ __forceinline__ float gelu_libm(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_exact(float x) {
#ifdef ISG_GELU_LIBM
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#else
  const float t = fminf(fabsf(x) * 0.70710678118654752440f, 6.0f);
  float r = 1.1604608516790904e-05f;
  r = fmaf(r, t, -0.00015296229685191065f);
  r = fmaf(r, t, 0.0008482258417643607f);
  r = fmaf(r, t, -0.002274767030030489f);
  r = fmaf(r, t, 8.478287054458633e-05f);
  r = fmaf(r, t, 0.02772449143230915f);
  r = fmaf(r, t, -0.1483079195022583f);
  r = fmaf(r, t, -0.9184429049491882f);
  r = fmaf(r, t, -1.6279072761535645f);
  const float e = __builtin_amdgcn_exp2f(r * t);
  return (0.5f * x) * (x > 0.f ? 2.0f - e : e);
#endif
}

// ---- Philox4x32-10 counter RNG (in-kernel noise when the caller passes none) ---------------------
struct Philox {
  static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  static __device__ __forceinline__ uint32_t draw(uint64_t seed, uint32_t a, uint32_t b) {
    uint32_t c[4] = {a, b, 0x1571u, 0x9E37u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    return c[0];
  }
};

// u in [0,1) with 24 random bits, then the reference's Uniform(tiny, 1-eps) -> Gumbel(loc, scale)
// transform chain (torch.distributions.Gumbel; ISubGVQA/sampling/methods/gumbel_scheme.py:65-69).
__device__ __forceinline__ float gumbel_from_bits(uint32_t bits, float loc, float scale) {
  float u01 = (float)(bits >> 8) * (1.0f / 16777216.0f);
  const float lo = FLT_MIN, hi = 1.0f - FLT_EPSILON;
  float u = add_rn(lo, mul_rn(u01, hi - lo));
  float l1 = (float)log((double)u);
  float l2 = (float)log((double)(-l1));
  return sub_rn(loc, mul_rn(scale, l2));
}

}  // namespace isg
