// Shared declarations of the two message-passing kernels (node-chunk form in isg_mp.hip, per-graph
// LDS-resident form in isg_mp_graph.hip).
#pragma once
#include "isg_common.hpp"

namespace isg {

constexpr int MP_WAVES = 4;
constexpr int MP_NPB = 16;     // destination nodes per workgroup (node-chunk kernel)
constexpr int MP_ECAP = 1024;  // CSR slots staged in LDS per workgroup (rest read from global)
constexpr int MP_LCAP = 32;    // logits per wave kept in LDS (x heads of the wave)

struct MpArgs {
  const float4 *x_l, *x_r, *e_proj, *att;
  const float4 *bias;     // optional
  const int *rowptr, *eid, *src;
  const float *node_mask, *edge_mask;        // optional (NULL: the layer is not masked)
  float4 *out;
  float *alpha;
  int N, C, H;
  int lde4;               // row stride of e_proj in float4 (H*C/4 when dense; L*H*C/4 when the layers' lin_edge
                          // projections are one fused [E, L*H*C] GEMM)
  int ldl4, ldr4;         // row stride of x_l / x_r in float4 (H*C/4 when dense; larger when they are column
                          // slices of one fused [N, 2*H*C] projection)
  float slope;
  const int *graph_ptr, *graph_eptr, *dst;   // per-graph kernel only (optional: NULL selects the node-chunk kernel)
  int B, lrows;           // graphs; x_l rows of a graph kept in LDS
  int f16;                // x_l / x_r / e_proj / out hold fp16 (per-graph kernel only)
  int flags;              // bit0: non-temporal e_proj loads / out stores; bit1: XCD-aware chunk mapping (chunk kernel);
                          // per-graph kernel: bit2 non-temporal x_l staging loads, bit5 x_r loads, bit6 alpha stores;
                          // bits 3 / 4 skip the logit / aggregation phase (ablation)
  int nchunks;
  const float *logits;    // per-graph grouped kernel: fp32 [E, H] attention logits in CSR SLOT order (isg_gatv2_edge_logits), or
                          // NULL.  When given, the logit phase (and with it e_proj and x_r) is skipped
  float *rowmax;          // per-graph kernel, fp32 rows: largest |out| per (node, head) [N, H], or NULL: the row scales of
                          // the fp16 three-product GEMM that consumes `out` (isg_linear_f16x3_tile) come from here
  uint16_t *planes;       // flat per-graph kernel (H = 4, two heads per workgroup), instead of `out` (or NULL): the result as the SEGMENTED
  float *planes_inv;      // planes32 operand of isg_linear_h3p -- columns [0, 2C) and [2C, 4C) each padded to whole 32-column
  int planes_kt;          // lines and each under its own row scale planes_inv[segment * N + node] (a workgroup owns two heads:
                          // it knows that half row's largest magnitude, not the whole row's); planes_kt = lines per row
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4 *p, bool nt) {
  if (!nt) return *p;
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(float4 *p, const float4 &v, bool nt) {
  if (nt) {
    f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(p));
  } else {
    *p = v;
  }
}

// ---- feature rows stored as fp32 (16 bytes per 4 channels) or fp16 (8 bytes per 4 channels; BASELINE configs[4]:
//      "fp16 features / fp32 accumulate") -- same index units (4 channels), the arithmetic is always fp32 ----
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 h4_to_f4(u32x2 r) {
  const f16x4 h = __builtin_bit_cast(f16x4, r);
  return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
}
__device__ __forceinline__ u32x2 f4_to_h4(const float4 &v) {
  f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};   // round to nearest even
  return __builtin_bit_cast(u32x2, h);
}
// raw register image of 4 channels (conversion is deferred to the point of use so that a batch of loads stays in flight)
template <bool F16> struct RawQ { typedef float4 type; };
template <> struct RawQ<true> { typedef u32x2 type; };
__device__ __forceinline__ float4 cvtq(const float4 &v) { return v; }
__device__ __forceinline__ float4 cvtq(const u32x2 &v) { return h4_to_f4(v); }
template <bool F16>
__device__ __forceinline__ typename RawQ<F16>::type ldraw(const float4 *base, size_t i) {
  if constexpr (F16) return reinterpret_cast<const u32x2 *>(base)[i];
  else return base[i];
}
template <bool F16>
__device__ __forceinline__ typename RawQ<F16>::type ldraw_stream(const float4 *base, size_t i, bool nt) {
  if constexpr (F16) {
    const u32x2 *p = reinterpret_cast<const u32x2 *>(base) + i;
    return nt ? __builtin_nontemporal_load(p) : *p;
  } else {
    return ld_stream(base + i, nt);
  }
}
template <bool F16>
__device__ __forceinline__ float4 ldq(const float4 *base, size_t i) {
  if constexpr (F16) return h4_to_f4(reinterpret_cast<const u32x2 *>(base)[i]);
  else return base[i];
}
template <bool F16>
__device__ __forceinline__ float4 ldq_stream(const float4 *base, size_t i, bool nt) {
  if constexpr (F16) {
    const u32x2 *p = reinterpret_cast<const u32x2 *>(base) + i;
    return h4_to_f4(nt ? __builtin_nontemporal_load(p) : *p);
  } else {
    return ld_stream(base + i, nt);
  }
}
template <bool F16>
__device__ __forceinline__ void stq_stream(float4 *base, size_t i, const float4 &v, bool nt) {
  if constexpr (F16) {
    u32x2 *p = reinterpret_cast<u32x2 *>(base) + i;
    const u32x2 h = f4_to_h4(v);
    if (nt) __builtin_nontemporal_store(h, p); else *p = h;
  } else {
    st_stream(base + i, v, nt);
  }
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// isg_mp_graph.hip: returns ISG_EUNSUPPORTED when the shape has no per-graph instantiation
int launch_mp_graph(MpArgs a, int nmax_host, int emax_host, hipStream_t st);

}  // namespace isg
