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
  const float4 *x_l, *x_r, *e_proj, *att, *bias;
  const int *rowptr, *eid, *src;
  const float *node_mask, *edge_mask;
  float4 *out;
  float *alpha;
  int N, C, H;
  int lde4;               // row stride of e_proj in float4 (H*C/4 when dense; L*H*C/4 when the layers' lin_edge
                          // projections are one fused [E, L*H*C] GEMM)
  int ldl4, ldr4;         // row stride of x_l / x_r in float4 (H*C/4 when dense; larger when they are column
                          // slices of one fused [N, 2*H*C] projection)
  float slope;
  const int *graph_ptr, *graph_eptr, *dst;   // per-graph kernel only
  int B, lrows;           // graphs; x_l rows of a graph kept in LDS
  int flags;              // bit0: non-temporal e_proj loads / out stores; bit1: XCD-aware chunk mapping
  int nchunks;
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

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// isg_mp_graph.hip: returns ISG_EUNSUPPORTED when the shape has no per-graph instantiation
int launch_mp_graph(MpArgs a, int nmax_host, int emax_host, hipStream_t st);

}  // namespace isg
