// GATv2 message passing, per-graph form: the x_l rows of one scene graph live in LDS.
//
// Why: in the node-chunk kernel (isg_mp.hip) every edge gathers its source row x_l[j] from L2 / Infinity Cache /
// HBM; PMC counters show ~1.5x the algorithmic read traffic there (profiles/r01_b_mp_traffic.md).  A PyG batch
// keeps a graph's nodes contiguous and its edges inside the graph, so one workgroup can stage the graph's x_l
// slice with coalesced loads ONCE and serve every x_l[j] (logit phase and aggregation phase) from LDS; HBM then
// sees exactly the algorithmic bytes (measured: FETCH 0.769 GB + WRITE 0.175 GB per launch at configs[1]).
//
//   grid  = graphs x (H / HS) items, head group fastest; a workgroup owns one graph and HS consecutive heads
//           (HS*C*4 <= ~1.25 KB per row keeps a 36-node graph at <= 45 KB of LDS, i.e. 3-4 workgroups per CU)
//   block = 8 waves
//   lanes = HS groups of G = 64/HS lanes (group = head), lane l owns float4 columns l, l+G, ... (P passes)
//   phase A  stage: x_l slice rows (a wave copies whole rows), one packed CSR record per slot
//            {source, edge id, destination, edge-mask value}, row pointers -> LDS                      (1 barrier)
//   phase B  EDGE-parallel logits: waves take blocks of U CSR slots; the U e_proj rows (streamed, non-temporal)
//            and U x_r rows are requested together, x_l[j] comes from LDS; per-head logit = G-lane DPP
//            butterfly -> LDS logit table                                                              (1 barrier)
//   phase C  NODE-parallel: a wave owns a destination node: max / exp-sum(+1e-16) / normalise over its segment of
//            the logit table, alpha out, aggregation of LDS-resident x_l rows in edge-id order, one fma per term,
//            + bias, row store.  No global loads in this phase.
// Ablation (profiles/r01_c_mp_ablation.md) showed the first version instruction-issue bound, not HBM bound
// (phases add up instead of overlapping, SIMDs ~95 % busy).  Hence: every wave-uniform value (slot record, row
// bases) is forced into SGPRs with readfirstlane, row addresses are scalar base + 32-bit lane offset, the CSR slot
// is one 16-byte LDS record, exp/reciprocal are the hardware instructions, and there is no fallback code in this
// kernel: it only runs when EVERY graph of the batch fits the CSR tables (host-checked from the plan's max nodes /
// max edges per graph; two table sizes are instantiated: 64 nodes / 256 edges and 256 nodes / 1024 edges); other
// batches use the node-chunk kernel.  The x_l window holds as many rows as fit 40 KB per
// workgroup; rows of a larger graph are read from global memory under a wave-uniform branch.  A source id outside
// its graph (never produced by PyG batching) is clamped into the graph instead of faulting.
#include "isg_mp.hpp"
#include "isg_f16x3.hpp"

#include <stdlib.h>

namespace isg {

constexpr int GK_WAVES = 8;
constexpr int GK_THREADS = GK_WAVES * 64;
// table sizes (nodes, CSR slots per graph) of the two instantiations: scene-graph sized batches, and skewed batches
// with hubs (BASELINE configs[4]: up to 200 nodes); the small one leaves more LDS for x_l rows
constexpr int GK_NCAP_S = 64, GK_ECAP_S = 256;
constexpr int GK_NCAP_L = 256, GK_ECAP_L = 1024;
#ifndef ISG_GK_U
#define ISG_GK_U 4
#endif
constexpr int GK_U = ISG_GK_U;
#ifndef ISG_GKF_U
#define ISG_GKF_U 1      // the flat kernel: ONE slot in flight per wave (62 VGPRs) and a 40 KB window, i.e. more workgroups per CU,
#endif                   // beat two slots (112 VGPRs) and a 78 KB window by 0.6 ms per full-model step (profiles/r04_ao_*)
// slots a wave has in flight in phase B (tuning builds: -DISG_GK_U=n)

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

template <int HS, int P, bool MASKED, bool EXACT, int GK_NCAP, int GK_ECAP, bool F16>
__global__ __launch_bounds__(GK_THREADS) void gatv2_mp_graph_kernel(MpArgs a) {
  constexpr int G = 64 / HS;
  extern __shared__ __attribute__((aligned(16))) float4 s_xl[];   // [lrows][HS*Q]
  __shared__ __attribute__((aligned(16))) int4 s_tab[GK_ECAP];    // {src - nb, eid, dst - nb, bits(edge mask)}
  __shared__ __attribute__((aligned(16))) float s_lg[GK_ECAP * HS];
  __shared__ __attribute__((aligned(16))) int s_rowptr[GK_NCAP + 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int nhg = a.H / HS;
  const int item = blockIdx.x;
  const int g = item / nhg, hg = item - g * nhg;
  // one round of scalar loads gives node range and CSR range (eptr[g] = rowptr[ptr[g]])
  const int nb = a.graph_ptr[g], n = min(a.graph_ptr[g + 1] - nb, GK_NCAP);
  const int e0 = a.graph_eptr[g], ne = min(a.graph_eptr[g + 1] - e0, GK_ECAP);
  if (n <= 0) return;
  const int Q = a.C >> 2;        // float4 per head
  const int R = a.H * Q;         // float4 per full row
  const int RQ = HS * Q;         // float4 per staged row slice
  const int hoff = hg * RQ;      // first float4 of this workgroup's head slice inside a row
  const int rows = min(n, a.lrows);
  const bool nt = a.flags & 1;
  const bool nt_xl = a.flags & 4;   // the staged x_l slice is read exactly once per workgroup
  const bool nt_xr = a.flags & 32, nt_al = a.flags & 64;   // experiments: x_r rows, alpha stores

  // ---- phase A: stage (loads first, LDS stores after) ------------------------------------------------------------
  {
    constexpr int RECS = (GK_ECAP + GK_THREADS - 1) / GK_THREADS;   // CSR records per thread
    int4 rec[RECS];
    int v_rp = 0;
    if (tid <= n) v_rp = a.rowptr[nb + tid] - e0;
#pragma unroll
    for (int k = 0; k < RECS; ++k) {
      const int t = tid + k * GK_THREADS;
      rec[k] = make_int4(0, 0, 0, __float_as_int(1.f));
      if (t < ne) {
        const int s = a.src[e0 + t], e = a.eid[e0 + t], d = a.dst[e0 + t];
        rec[k].x = min(max(s - nb, 0), n - 1);   // a source outside its graph is clamped into it, never out of bounds
        rec[k].y = e;
        rec[k].z = d - nb;
        if (MASKED) {
          float me;
          if (a.edge_mask) me = a.edge_mask[e];
          else me = a.node_mask[s] * a.node_mask[d];    // NodeMaskToEdgeMask, fused
          rec[k].w = __float_as_int(me);
        }
      }
    }
    // x_l slice: wave w copies rows w, w+8, ...; a lane covers columns lane, lane+64, ... of the slice
#pragma unroll 1
    for (int r0 = wave; r0 < rows; r0 += GK_WAVES * 4) {
#pragma unroll 1
      for (int c = lane; c < RQ; c += 64) {
        typename RawQ<F16>::type v0, v1, v2, v3;
        const int r1 = r0 + GK_WAVES, r2 = r0 + 2 * GK_WAVES, r3 = r0 + 3 * GK_WAVES;
        v0 = ldraw_stream<F16>(a.x_l, (size_t)(nb + r0) * a.ldl4 + hoff + c, nt_xl);
        if (r1 < rows) v1 = ldraw_stream<F16>(a.x_l, (size_t)(nb + r1) * a.ldl4 + hoff + c, nt_xl);
        if (r2 < rows) v2 = ldraw_stream<F16>(a.x_l, (size_t)(nb + r2) * a.ldl4 + hoff + c, nt_xl);
        if (r3 < rows) v3 = ldraw_stream<F16>(a.x_l, (size_t)(nb + r3) * a.ldl4 + hoff + c, nt_xl);
        s_xl[r0 * RQ + c] = cvtq(v0);
        if (r1 < rows) s_xl[r1 * RQ + c] = cvtq(v1);
        if (r2 < rows) s_xl[r2 * RQ + c] = cvtq(v2);
        if (r3 < rows) s_xl[r3 * RQ + c] = cvtq(v3);
      }
    }
    if (tid <= n) s_rowptr[tid] = v_rp;
#pragma unroll
    for (int k = 0; k < RECS; ++k)
      if (tid + k * GK_THREADS < ne) s_tab[tid + k * GK_THREADS] = rec[k];
    if (a.logits)     // logits formed by isg_gatv2_edge_logits (slot order): this workgroup's head slice of its graph's slots
      for (int t = tid; t < ne * HS; t += GK_THREADS)
        s_lg[t] = a.logits[(size_t)(e0 + t / HS) * a.H + hg * HS + (t % HS)];
  }
  __syncthreads();

  const int grp = lane / G, l = lane % G;
  int off[P];      // float4 offset inside the slice
  bool ok[P];
  float4 att4[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int q = p * G + l;
    ok[p] = EXACT || q < Q;
    off[p] = grp * Q + (ok[p] ? q : 0);
    att4[p] = ok[p] ? a.att[hoff + off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int hd = hg * HS + grp;   // global head index of this lane group
  const float slope = a.slope;

  // ---- phase B: edge-parallel logits (skipped when the logits were handed in) ----------------------------------------
  if (!(a.flags & 8) && !a.logits) {
#pragma unroll 1
    for (int tb = wave * GK_U; tb < ne; tb += GK_WAVES * GK_U) {
      typename RawQ<F16>::type epv[GK_U][P], xrv[GK_U][P];
      int jl[GK_U];
      float me[GK_U];
#pragma unroll
      for (int u = 0; u < GK_U; ++u) {
        const int t = min(tb + u, ne - 1);          // tail slots repeat the last edge (their result is dropped)
        const int4 rec = s_tab[t];
        jl[u] = uni(rec.x);
        me[u] = unif(__int_as_float(rec.w));
        const size_t ep = (size_t)uni(rec.y) * a.lde4 + hoff;
        const size_t xr = (size_t)(nb + uni(rec.z)) * a.ldr4 + hoff;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (ok[p]) {
            epv[u][p] = ldraw_stream<F16>(a.e_proj, ep + off[p], nt);
            xrv[u][p] = ldraw_stream<F16>(a.x_r, xr + off[p], nt_xr);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < GK_U; ++u) {
        const int t = tb + u;
        // rows of a graph larger than the LDS window (rare: the window is sized for 4 workgroups per CU) come
        // from global memory; LDS read from a clamped row first, override under a wave-uniform branch
        const bool in_lds = jl[u] < rows;
        const float4 *xl_s = s_xl + (in_lds ? jl[u] : 0) * RQ;
        float part = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (ok[p]) {
            const float4 v = cvtq(epv[u][p]), r4 = cvtq(xrv[u][p]);
            float4 w4 = xl_s[off[p]];
            if (!in_lds) w4 = ldq<F16>(a.x_l, (size_t)(nb + jl[u]) * a.ldl4 + hoff + off[p]);
            float4 s;
            s.x = (r4.x + w4.x) + v.x;
            s.y = (r4.y + w4.y) + v.y;
            s.z = (r4.z + w4.z) + v.z;
            s.w = (r4.w + w4.w) + v.w;
            if (MASKED) { s.x *= me[u]; s.y *= me[u]; s.z *= me[u]; s.w *= me[u]; }
            s.x = leaky(s.x, slope); s.y = leaky(s.y, slope); s.z = leaky(s.z, slope); s.w = leaky(s.w, slope);
            if (MASKED) { s.x *= me[u]; s.y *= me[u]; s.z *= me[u]; s.w *= me[u]; }
            part += dot4(s, att4[p]);
          }
        }
        const float logit = group_sum<G>(part);
        if (l == 0 && t < ne) s_lg[t * HS + grp] = logit;
      }
    }
  }
  __syncthreads();

  // ---- phase C: node-parallel softmax + aggregation; every operand in LDS ------------------------------------------
  if (!(a.flags & 16)) {
#pragma unroll 1
    for (int k = wave; k < n; k += GK_WAVES) {
      const int rb = uni(s_rowptr[k]), re = min(uni(s_rowptr[k + 1]), ne);
      float mx = -INFINITY;
#pragma unroll 1
      for (int t = rb; t < re; ++t) mx = fmaxf(mx, s_lg[t * HS + grp]);
      float den = 0.f;
#pragma unroll 1
      for (int t = rb; t < re; ++t) den += __builtin_amdgcn_exp2f((s_lg[t * HS + grp] - mx) * 1.4426950408889634f);
      const float rden = __builtin_amdgcn_rcpf(den + 1e-16f);
      float4 acc[P];
#pragma unroll
      for (int p = 0; p < P; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
      for (int t = rb; t < re; ++t) {
        const int4 rec = s_tab[t];
        const float w = __builtin_amdgcn_exp2f((s_lg[t * HS + grp] - mx) * 1.4426950408889634f) * rden;
        if (l == 0) {
          if (nt_al) __builtin_nontemporal_store(w, a.alpha + (size_t)uni(rec.y) * a.H + hd);
          else a.alpha[(size_t)uni(rec.y) * a.H + hd] = w;
        }
        const float wm = MASKED ? mul_rn(w, unif(__int_as_float(rec.w))) : w;
        const int jl = uni(rec.x);
        const bool in_lds = jl < rows;
        const float4 *xl_s = s_xl + (in_lds ? jl : 0) * RQ;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (ok[p]) {
            float4 u4 = xl_s[off[p]];
            if (!in_lds) u4 = ldq<F16>(a.x_l, (size_t)(nb + jl) * a.ldl4 + hoff + off[p]);
            acc[p].x = fmaf(u4.x, wm, acc[p].x);
            acc[p].y = fmaf(u4.y, wm, acc[p].y);
            acc[p].z = fmaf(u4.z, wm, acc[p].z);
            acc[p].w = fmaf(u4.w, wm, acc[p].w);
          }
        }
      }
      const size_t orow = (size_t)(nb + k) * R + hoff;
      float rmx = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          float4 o = acc[p];
          if (a.bias) {
            const float4 b = a.bias[hoff + off[p]];
            o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
          }
          rmx = fmaxf(rmx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
          stq_stream<F16>(a.out, orow + off[p], o, nt);
        }
      }
      if (!F16 && a.rowmax) {   // largest magnitude of this (node, head): the consumer GEMM's row scale
        rmx = group_max<G>(rmx);
        if (l == 0) a.rowmax[(size_t)(nb + k) * a.H + hd] = rmx;
      }
    }
  }
}


// ---- FLAT lane mapping: head dimensions that do not tile 64 lanes (the reference's default C = 300: 75 float4 per head) --
// The grouped kernel above gives every head its own lane group of 64 / HS lanes; at C = 300 that is one head per
// workgroup in two passes, the second mostly idle (75 of 128 lane slots), and the CSR records staged once per HEAD:
// 762 us at H*C = 1200 on the configs[1] topology, worse than the node-chunk kernel (672 us, 1.41x traffic).  Here a
// workgroup owns HS heads (HS * Q float4 per row slice) and its lanes walk the slice FLAT, lane l -> float4 l, l + 64, ...
// (P passes; 150 of 192 slots at HS = 2, C = 300); a pass's element belongs to head (index / Q), so a lane may serve two
// heads: per-head logits are HS full-wave sums of per-head partials, and the softmax weights of all HS heads are formed
// in every lane.  Phases, LDS tables, arithmetic and summation orders are those of the grouped kernel.
template <int HS, int P, bool MASKED, int GK_NCAP, int GK_ECAP>
__global__ __launch_bounds__(GK_THREADS) void gatv2_mp_graph_flat_kernel(MpArgs a) {
  constexpr int U = ISG_GKF_U;    // CSR slots a wave has in flight in phase B (P passes x 2 operands x U rows of loads)
  extern __shared__ __attribute__((aligned(16))) float4 s_xl[];   // [lrows][HS*Q]
  __shared__ __attribute__((aligned(16))) int4 s_tab[GK_ECAP];    // {src - nb, eid, dst - nb, bits(edge mask)}
  __shared__ __attribute__((aligned(16))) float s_lg[GK_ECAP * HS];
  __shared__ __attribute__((aligned(16))) int s_rowptr[GK_NCAP + 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int nhg = a.H / HS;
  const int item = blockIdx.x;
  const int g = item / nhg, hg = item - g * nhg;
  const int nb = a.graph_ptr[g], n = min(a.graph_ptr[g + 1] - nb, GK_NCAP);
  const int e0 = a.graph_eptr[g], ne = min(a.graph_eptr[g + 1] - e0, GK_ECAP);
  if (n <= 0) return;
  const int Q = a.C >> 2, R = a.H * Q, RQ = HS * Q, hoff = hg * RQ;
  const int rows = min(n, a.lrows);
  const bool nt = a.flags & 1, nt_xl = a.flags & 4;

  // ---- phase A: stage the x_l slice, the CSR records and the row pointers ------------------------------------------
  {
    constexpr int RECS = (GK_ECAP + GK_THREADS - 1) / GK_THREADS;
    int4 rec[RECS];
    int v_rp = 0;
    if (tid <= n) v_rp = a.rowptr[nb + tid] - e0;
#pragma unroll
    for (int k = 0; k < RECS; ++k) {
      const int t = tid + k * GK_THREADS;
      rec[k] = make_int4(0, 0, 0, __float_as_int(1.f));
      if (t < ne) {
        const int s = a.src[e0 + t], e = a.eid[e0 + t], d = a.dst[e0 + t];
        rec[k].x = min(max(s - nb, 0), n - 1);
        rec[k].y = e;
        rec[k].z = d - nb;
        if (MASKED) {
          float me;
          if (a.edge_mask) me = a.edge_mask[e];
          else me = a.node_mask[s] * a.node_mask[d];
          rec[k].w = __float_as_int(me);
        }
      }
    }
#pragma unroll 1
    for (int r0 = wave; r0 < rows; r0 += GK_WAVES * 2) {
      const int r1 = r0 + GK_WAVES;
#pragma unroll 1
      for (int c = lane; c < RQ; c += 64) {
        float4 v0, v1;
        v0 = ld_stream(a.x_l + (size_t)(nb + r0) * a.ldl4 + hoff + c, nt_xl);
        if (r1 < rows) v1 = ld_stream(a.x_l + (size_t)(nb + r1) * a.ldl4 + hoff + c, nt_xl);
        s_xl[r0 * RQ + c] = v0;
        if (r1 < rows) s_xl[r1 * RQ + c] = v1;
      }
    }
    if (tid <= n) s_rowptr[tid] = v_rp;
#pragma unroll
    for (int k = 0; k < RECS; ++k)
      if (tid + k * GK_THREADS < ne) s_tab[tid + k * GK_THREADS] = rec[k];
    if (a.logits)     // logits formed by isg_gatv2_edge_logits (slot order): this workgroup's HS heads of its graph's slots
      for (int t = tid; t < ne * HS; t += GK_THREADS)
        s_lg[t] = a.logits[(size_t)(e0 + t / HS) * a.H + hg * HS + (t % HS)];
  }
  __syncthreads();

  int off[P], hof[P];     // float4 offset inside the slice; head (0 .. HS-1) it belongs to
  bool ok[P];
  float4 att4[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane + 64 * p;
    ok[p] = i < RQ;
    off[p] = ok[p] ? i : 0;
    hof[p] = off[p] / Q;
    att4[p] = ok[p] ? a.att[hoff + off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float slope = a.slope;

  // ---- phase B: edge-parallel logits (skipped when the logits were handed in: neither e_proj nor x_r is touched) ----------
#pragma unroll 1
  for (int tb = a.logits ? ne : wave * U; tb < ne; tb += GK_WAVES * U) {
    float4 epv[U][P], xrv[U][P];
    int jl[U];
    float me[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min(tb + u, ne - 1);
      const int4 rec = s_tab[t];
      jl[u] = uni(rec.x);
      me[u] = unif(__int_as_float(rec.w));
      const size_t ep = (size_t)uni(rec.y) * a.lde4 + hoff;
      const size_t xr = (size_t)(nb + uni(rec.z)) * a.ldr4 + hoff;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          epv[u][p] = ld_stream(a.e_proj + ep + off[p], nt);
          xrv[u][p] = a.x_r[xr + off[p]];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + u;
      const bool in_lds = jl[u] < rows;
      const float4 *xl_s = s_xl + (in_lds ? jl[u] : 0) * RQ;
      float part[HS];
#pragma unroll
      for (int hh = 0; hh < HS; ++hh) part[hh] = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          const float4 v = epv[u][p], r4 = xrv[u][p];
          float4 w4 = xl_s[off[p]];
          if (!in_lds) w4 = a.x_l[(size_t)(nb + jl[u]) * a.ldl4 + hoff + off[p]];
          float4 s;
          s.x = (r4.x + w4.x) + v.x;
          s.y = (r4.y + w4.y) + v.y;
          s.z = (r4.z + w4.z) + v.z;
          s.w = (r4.w + w4.w) + v.w;
          if (MASKED) { s.x *= me[u]; s.y *= me[u]; s.z *= me[u]; s.w *= me[u]; }
          s.x = leaky(s.x, slope); s.y = leaky(s.y, slope); s.z = leaky(s.z, slope); s.w = leaky(s.w, slope);
          if (MASKED) { s.x *= me[u]; s.y *= me[u]; s.z *= me[u]; s.w *= me[u]; }
          const float d = own_reg(dot4(s, att4[p]));     // a scalar: the head select below must not take it from a pair's high dword
#pragma unroll
          for (int hh = 0; hh < HS; ++hh) part[hh] += hof[p] == hh ? d : 0.f;
        }
      }
#pragma unroll
      for (int hh = 0; hh < HS; ++hh) {
        const float logit = wave_sum(part[hh]);
        if (lane == 0 && t < ne) s_lg[t * HS + hh] = logit;
      }
    }
  }
  __syncthreads();

  // ---- phase C: node-parallel softmax + aggregation; every operand in LDS ---------------------------------------------
#pragma unroll 1
  for (int k = wave; k < n; k += GK_WAVES) {
    const int rb = uni(s_rowptr[k]), re = min(uni(s_rowptr[k + 1]), ne);
    float mx[HS], rden[HS];
#pragma unroll
    for (int hh = 0; hh < HS; ++hh) {
      float m = -INFINITY;
#pragma unroll 1
      for (int t = rb; t < re; ++t) m = fmaxf(m, s_lg[t * HS + hh]);
      float den = 0.f;
#pragma unroll 1
      for (int t = rb; t < re; ++t) den += __builtin_amdgcn_exp2f((s_lg[t * HS + hh] - m) * 1.4426950408889634f);
      mx[hh] = m;
      rden[hh] = __builtin_amdgcn_rcpf(den + 1e-16f);
    }
    float4 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int t = rb; t < re; ++t) {
      const int4 rec = s_tab[t];
      float w[HS];
#pragma unroll
      for (int hh = 0; hh < HS; ++hh) {
        w[hh] = __builtin_amdgcn_exp2f((s_lg[t * HS + hh] - mx[hh]) * 1.4426950408889634f) * rden[hh];
        if (lane == 0) a.alpha[(size_t)uni(rec.y) * a.H + hg * HS + hh] = w[hh];
        if (MASKED) w[hh] = mul_rn(w[hh], unif(__int_as_float(rec.w)));
      }
      const int jl = uni(rec.x);
      const bool in_lds = jl < rows;
      const float4 *xl_s = s_xl + (in_lds ? jl : 0) * RQ;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          float4 u4 = xl_s[off[p]];
          if (!in_lds) u4 = a.x_l[(size_t)(nb + jl) * a.ldl4 + hoff + off[p]];
          float wm = w[0];
#pragma unroll
          for (int hh = 1; hh < HS; ++hh) wm = hof[p] == hh ? w[hh] : wm;
          acc[p].x = fmaf(u4.x, wm, acc[p].x);
          acc[p].y = fmaf(u4.y, wm, acc[p].y);
          acc[p].z = fmaf(u4.z, wm, acc[p].z);
          acc[p].w = fmaf(u4.w, wm, acc[p].w);
        }
      }
    }
    const size_t orow = (size_t)(nb + k) * R + hoff;
    if (a.planes) {
      // the half row (this workgroup's two heads) as planes32 lines under its own scale: x_proj.0 reads the result from
      // these, with no fp32 copy and no split pass in between (isg_linear_h3p's segmented A operand)
      float mx = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          if (a.bias) {
            const float4 b = a.bias[hoff + off[p]];
            acc[p].x += b.x; acc[p].y += b.y; acc[p].z += b.z; acc[p].w += b.w;
          }
          mx = fmaxf(mx, fmaxf(fmaxf(fabsf(acc[p].x), fabsf(acc[p].y)), fmaxf(fabsf(acc[p].z), fabsf(acc[p].w))));
        }
      }
      mx = wave_max(mx);
      float sc, inv;
      h3_scale(mx, sc, inv);
      if (lane == 0) a.planes_inv[(size_t)hg * a.N + nb + k] = inv;
      const int seg_kt = (RQ + 7) >> 3;
      _Float16 *pl = reinterpret_cast<_Float16 *>(a.planes) + ((size_t)(nb + k) * a.planes_kt + (size_t)hg * seg_kt) * 64;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = lane + 64 * p;
        if (i < seg_kt * 8) {                    // beyond the half row: the zeros of its k padding
          float4 t = ok[p] ? acc[p] : make_float4(0.f, 0.f, 0.f, 0.f);
          t.x *= sc; t.y *= sc; t.z *= sc; t.w *= sc;
          const hf16x4 hi = {(_Float16)t.x, (_Float16)t.y, (_Float16)t.z, (_Float16)t.w};
          const hf16x4 mid = {(_Float16)(t.x - (float)hi[0]), (_Float16)(t.y - (float)hi[1]), (_Float16)(t.z - (float)hi[2]),
                              (_Float16)(t.w - (float)hi[3])};
          _Float16 *d = pl + (i >> 3) * 64 + (i & 7) * 4;
          *reinterpret_cast<hf16x4 *>(d) = hi;
          *reinterpret_cast<hf16x4 *>(d + 32) = mid;
        }
      }
      continue;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (ok[p]) {
        float4 o = acc[p];
        if (a.bias) {
          const float4 b = a.bias[hoff + off[p]];
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        st_stream(a.out + orow + off[p], o, nt);
      }
    }
  }
}

template <int HS, int P, int NC, int EC>
static int launch_flat_sized(MpArgs a, int nmax_host, hipStream_t st) {
  const long long items = (long long)a.B * (a.H / HS);
  if (items >= (1ll << 31)) return ISG_EUNSUPPORTED;
  const size_t row_bytes = (size_t)HS * a.C * 4;
  const size_t static_bytes = (size_t)EC * 16 + (size_t)EC * HS * 4 + (NC + 4) * 4;
  // window: 40 KB = a 14-row window at H * C = 1200 and three to four workgroups per CU; a graph's rows beyond it are read from
  // global memory (L2).  ISG_MPF_LDS_KB: sweep switch (78 = two workgroups per CU with a 30-row window: the round-2 setting)
  static const int flat_kb = [] { const char *e = getenv("ISG_MPF_LDS_KB"); const int v = e ? atoi(e) : 40; return v < 16 || v > 78 ? 40 : v; }();
  const size_t budget = (size_t)flat_kb * 1024;
  if (budget < static_bytes + 8 * row_bytes) return ISG_EUNSUPPORTED;
  a.lrows = (int)((budget - static_bytes) / row_bytes);
  if (a.lrows > nmax_host) a.lrows = nmax_host;
  const size_t dyn = (size_t)a.lrows * row_bytes;
  dim3 grid((unsigned)items), block(GK_THREADS);
  const bool masked = a.node_mask || a.edge_mask;
  if (masked) {
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&gatv2_mp_graph_flat_kernel<HS, P, true, NC, EC>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    if (!ok) return ISG_EUNSUPPORTED;
    gatv2_mp_graph_flat_kernel<HS, P, true, NC, EC><<<grid, block, dyn, st>>>(a);
  } else {
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&gatv2_mp_graph_flat_kernel<HS, P, false, NC, EC>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    if (!ok) return ISG_EUNSUPPORTED;
    gatv2_mp_graph_flat_kernel<HS, P, false, NC, EC><<<grid, block, dyn, st>>>(a);
  }
  return check_launch();
}

// fp32 rows, scene-graph sized graphs (64 nodes / 256 edges): HS heads per workgroup, P = ceil(HS * C / 256) passes
static int launch_mp_graph_flat(const MpArgs &a, int nmax_host, int emax_host, hipStream_t st) {
  if (a.f16 || nmax_host > GK_NCAP_S || emax_host > GK_ECAP_S || (a.H & 1)) return ISG_EUNSUPPORTED;
  if (a.planes && a.H != 4) return ISG_EUNSUPPORTED;      // two segments = two workgroups of two heads each
  const int RQ = 2 * (a.C >> 2);
  const int P = (RQ + 63) / 64;
  if (P == 3) return launch_flat_sized<2, 3, GK_NCAP_S, GK_ECAP_S>(a, nmax_host, st);
  if (P == 2) return launch_flat_sized<2, 2, GK_NCAP_S, GK_ECAP_S>(a, nmax_host, st);
  if (P == 4) return launch_flat_sized<2, 4, GK_NCAP_S, GK_ECAP_S>(a, nmax_host, st);
  return ISG_EUNSUPPORTED;
}

template <int HS, int P, int NC, int EC>
static int launch_sized(MpArgs a, int nmax_host, hipStream_t st) {
  const long long items = (long long)a.B * (a.H / HS);
  if (items >= (1ll << 31)) return ISG_EUNSUPPORTED;
  const size_t row_bytes = (size_t)HS * a.C * 4;
  const size_t static_bytes = (size_t)EC * 16 + (size_t)EC * HS * 4 + (NC + 4) * 4;
  // LDS window: as many x_l rows as fit 40 KB per workgroup (4 workgroups = 32 waves per CU), never more than the
  // largest graph needs; rows beyond the window are read from global memory
  static const int lds_kb = [] { const char *e = getenv("ISG_MP_LDS_KB"); return e ? atoi(e) : 40; }();   // read once, at first use
  const size_t budget = (size_t)lds_kb * 1024;
  if (budget < static_bytes + 8 * row_bytes) return ISG_EUNSUPPORTED;
  a.lrows = (int)((budget - static_bytes) / row_bytes);
  if (a.lrows > nmax_host) a.lrows = nmax_host;
  const size_t dyn = (size_t)a.lrows * row_bytes;
  dim3 grid((unsigned)items), block(GK_THREADS);
  const bool masked = a.node_mask || a.edge_mask;
  const bool exact = (a.C >> 2) == (64 / HS) * P;
#define ISG_GK_LAUNCH(M_, X_, F_) gatv2_mp_graph_kernel<HS, P, M_, X_, NC, EC, F_><<<grid, block, dyn, st>>>(a)
  if (a.f16) {
    if (masked) { if (exact) ISG_GK_LAUNCH(true, true, true); else ISG_GK_LAUNCH(true, false, true); }
    else { if (exact) ISG_GK_LAUNCH(false, true, true); else ISG_GK_LAUNCH(false, false, true); }
  } else {
    if (masked) { if (exact) ISG_GK_LAUNCH(true, true, false); else ISG_GK_LAUNCH(true, false, false); }
    else { if (exact) ISG_GK_LAUNCH(false, true, false); else ISG_GK_LAUNCH(false, false, false); }
  }
#undef ISG_GK_LAUNCH
  return check_launch();
}

template <int HS, int P>
static int launch_one(const MpArgs &a, int nmax_host, int emax_host, hipStream_t st) {
  if (nmax_host <= GK_NCAP_S && emax_host <= GK_ECAP_S) return launch_sized<HS, P, GK_NCAP_S, GK_ECAP_S>(a, nmax_host, st);
  return launch_sized<HS, P, GK_NCAP_L, GK_ECAP_L>(a, nmax_host, st);
}

// Runs only when every graph of the batch fits the kernel's LDS tables; ISG_EUNSUPPORTED otherwise (the caller
// then uses the node-chunk kernel).
int launch_mp_graph(MpArgs a, int nmax_host, int emax_host, hipStream_t st) {
  const int Q = a.C >> 2;
  if (nmax_host <= 0 || nmax_host > GK_NCAP_L || emax_host < 0 || emax_host > GK_ECAP_L) return ISG_EUNSUPPORTED;
  // heads per workgroup: the largest HS | H with a row slice of at most 1280 bytes (at least one head)
  int HS = 1;
  for (int hs = 8; hs >= 1; hs >>= 1)
    if (a.H % hs == 0 && hs * a.C * 4 <= 1280) { HS = hs; break; }
  const int G = 64 / HS;
  const int P = (Q + G - 1) / G;
  if (P > 2) return ISG_EUNSUPPORTED;
  // one head per workgroup whose row needs a second, mostly idle pass (the reference's default C = 300: 75 of 128 lane
  // slots, and the CSR tables staged once per HEAD): the node-chunk kernel is faster there (tools/time_mp_c300.py:
  // 672 vs 762 us at H = 4, C = 300 on the configs[1] topology)
  static const bool force_graph = getenv("ISG_MP_FORCE_GRAPH") != nullptr;   // read once (A/B switches)
  static const bool no_flat = getenv("ISG_MP_NO_FLAT") != nullptr;
  if (HS == 1 && a.H > 1 && P == 2 && Q * 10 < G * P * 7 && !force_graph) {
    // head dimension that fills < 70 % of two passes (the reference's C = 300): flat lane mapping over two heads
    if (!no_flat && !a.rowmax) {
      const int rc = launch_mp_graph_flat(a, nmax_host, emax_host, st);
      if (rc != ISG_EUNSUPPORTED) return rc;
    }
    return ISG_EUNSUPPORTED;      // node-chunk kernel
  }
  if (a.planes) return ISG_EUNSUPPORTED;      // the segmented planes32 result exists in the flat kernel only
#define ISG_GK(hs, p) if (HS == hs && P == p) return launch_one<hs, p>(a, nmax_host, emax_host, st)
  ISG_GK(1, 1); ISG_GK(1, 2);
  ISG_GK(2, 1); ISG_GK(2, 2);
  ISG_GK(4, 1); ISG_GK(4, 2);
  ISG_GK(8, 1); ISG_GK(8, 2);
#undef ISG_GK
  return ISG_EUNSUPPORTED;
}

}  // namespace isg
