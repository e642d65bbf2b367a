// GATv2 message passing, per-graph form: the x_l rows of one scene graph live in LDS.
//
// Why: in the node-chunk kernel (isg_mp.hip) every edge gathers its source row x_l[j] from L2 / Infinity Cache /
// HBM; PMC counters show ~1.5x the algorithmic read traffic there (profiles/r01_b_mp_traffic.md).  A PyG batch
// keeps a graph's nodes contiguous and its edges inside the graph, so one workgroup can stage the graph's x_l
// slice with coalesced loads ONCE and serve every x_l[j] (logit phase and aggregation phase) from LDS; HBM then
// sees exactly the algorithmic bytes (measured: FETCH 0.769 GB + WRITE 0.175 GB per launch at configs[1]).
//
//   grid  = graphs x (H / HS) items, head group fastest; a workgroup owns one graph and HS consecutive heads
//           (HS*C*4 <= ~1.25 KB per row keeps a 30-node graph at ~30 KB of LDS, i.e. 4 workgroups per CU)
//   block = 8 waves
//   lanes = HS groups of G = 64/HS lanes (group = head), lane l owns float4 columns l, l+G, ... (P passes)
//   phase A  stage: x_l slice rows (a wave copies whole rows: no index division), the graph's CSR (src, eid, dst,
//            rowptr) and node mask -> LDS; all loads issued before the first LDS store                 (1 barrier)
//   phase B  EDGE-parallel logits: waves take blocks of U CSR slots; the U e_proj rows (streamed, non-temporal)
//            and U x_r rows are requested together, x_l[j] comes from LDS; per-head logit = G-lane DPP
//            butterfly -> LDS logit table                                                              (1 barrier)
//   phase C  NODE-parallel: a wave owns a destination node: max / exp-sum(+1e-16) / divide over its segment of the
//            logit table, alpha out, aggregation of LDS-resident x_l rows in edge-id order with unfused mul+add,
//            + bias, row store.  No global loads in this phase.
// The kernel is VALU-issue bound before it is HBM bound (PMC: SIMDs ~95 % busy in the first version), so everything
// wave-uniform (edge id, source, destination, row bases) is forced into SGPRs with readfirstlane and row addresses
// are scalar base + 32-bit lane offset.
// Graphs that exceed the LDS tables (n > GK_NCAP nodes or > GK_ECAP edges) take the generic instantiation of the
// phases (tables fall back to global memory slot by slot), and a source row outside the staged window (or outside
// the graph) is read from global memory, so correctness never depends on the batch layout.
#include "isg_mp.hpp"

#include <stdlib.h>

namespace isg {

constexpr int GK_WAVES = 8;
constexpr int GK_THREADS = GK_WAVES * 64;
constexpr int GK_NCAP = 128;   // nodes of a graph the LDS tables hold (fast path)
constexpr int GK_ECAP = 256;   // CSR slots of a graph the LDS tables hold (fast path)

template <int HS>
struct GkShared {
  int rowptr[GK_NCAP + 4];
  int src[GK_ECAP];
  int eid[GK_ECAP];
  int dst[GK_ECAP];
  float nm[GK_NCAP];
  float lg[GK_ECAP * HS];
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

template <int HS, int P, int U>
__global__ __launch_bounds__(GK_THREADS) void gatv2_mp_graph_kernel(MpArgs a) {
  constexpr int G = 64 / HS;
  extern __shared__ __attribute__((aligned(16))) float4 s_xl[];   // [lrows][HS*Q]
  __shared__ __attribute__((aligned(16))) GkShared<HS> sh;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int nhg = a.H / HS;
  const int item = blockIdx.x;
  const int g = item / nhg, hg = item - g * nhg;
  // one round of scalar loads gives node range and CSR range (eptr[g] = rowptr[ptr[g]])
  const int nb = a.graph_ptr[g], n = a.graph_ptr[g + 1] - nb;
  const int e0 = a.graph_eptr[g], ne = a.graph_eptr[g + 1] - e0;
  if (n <= 0) return;
  const int Q = a.C >> 2;        // float4 per head
  const int R = a.H * Q;         // float4 per full row
  const int RQ = HS * Q;         // float4 per staged row slice
  const int hoff = hg * RQ;      // first float4 of this workgroup's head slice inside a row
  const int rows = min(n, a.lrows);
  const int mode = a.edge_mask ? 2 : (a.node_mask ? 1 : 0);
  const bool nt = a.flags & 1;
  const bool fast = n <= GK_NCAP && ne <= GK_ECAP;

  // ---- phase A: stage (loads first, LDS stores after) ------------------------------------------------------------
  {
    const int ncap = min(n, GK_NCAP), nes = min(ne, GK_ECAP);
    int v_rp = 0, v_src = 0, v_eid = 0, v_dst = 0;
    float v_nm = 1.f;
    if (tid <= ncap) v_rp = a.rowptr[nb + tid];
    if (tid < nes) {
      v_src = a.src[e0 + tid];
      v_eid = a.eid[e0 + tid];
      v_dst = a.dst[e0 + tid] - nb;
    }
    if (mode == 1 && tid < ncap) v_nm = a.node_mask[nb + tid];
    // x_l slice: wave w copies rows w, w+8, ...; a lane covers columns lane, lane+64, ... of the slice
    // (four rows in flight per wave; named registers, a local array here ends up in scratch)
#pragma unroll 1
    for (int r0 = wave; r0 < ((a.flags & 32) ? 0 : rows); r0 += GK_WAVES * 4) {
#pragma unroll 1
      for (int c = lane; c < RQ; c += 64) {
        float4 v0, v1, v2, v3;
        const int r1 = r0 + GK_WAVES, r2 = r0 + 2 * GK_WAVES, r3 = r0 + 3 * GK_WAVES;
        v0 = a.x_l[(size_t)(nb + r0) * R + hoff + c];
        if (r1 < rows) v1 = a.x_l[(size_t)(nb + r1) * R + hoff + c];
        if (r2 < rows) v2 = a.x_l[(size_t)(nb + r2) * R + hoff + c];
        if (r3 < rows) v3 = a.x_l[(size_t)(nb + r3) * R + hoff + c];
        s_xl[r0 * RQ + c] = v0;
        if (r1 < rows) s_xl[r1 * RQ + c] = v1;
        if (r2 < rows) s_xl[r2 * RQ + c] = v2;
        if (r3 < rows) s_xl[r3 * RQ + c] = v3;
      }
    }
    if (tid <= ncap) sh.rowptr[tid] = v_rp;
    if (tid < nes) { sh.src[tid] = v_src; sh.eid[tid] = v_eid; sh.dst[tid] = v_dst; }
    if (mode == 1 && tid < ncap) sh.nm[tid] = v_nm;
  }
  __syncthreads();

  const int grp = lane / G, l = lane % G;
  int off[P];      // float4 offset inside the slice
  bool ok[P];
  float4 att4[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int q = p * G + l;
    ok[p] = q < Q;
    off[p] = grp * Q + (ok[p] ? q : 0);
    att4[p] = ok[p] ? a.att[hoff + off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int hd = hg * HS + grp;   // global head index of this lane group
  const float slope = a.slope;

  // table accessors: with F (fast graph) everything is in LDS; otherwise slot-by-slot fallback to global memory.
  // The index t is wave-uniform, so the values are too: readfirstlane moves them (and all address math) to SGPRs.
  // (explicit if/else, never a ?: between an LDS and a global address: that would become a flat access)
#define A_SEL(T, COND, LDSV, GLBV) ({ T v_; if (COND) v_ = (LDSV); else v_ = (GLBV); v_; })
#define A_SRC(F, t) uni(A_SEL(int, (F) || (t) < GK_ECAP, sh.src[t], a.src[e0 + (t)]))
#define A_EID(F, t) uni(A_SEL(int, (F) || (t) < GK_ECAP, sh.eid[t], a.eid[e0 + (t)]))
#define A_DST(F, t) uni(A_SEL(int, (F) || (t) < GK_ECAP, sh.dst[t], a.dst[e0 + (t)] - nb))
#define A_RP(F, k) (uni(A_SEL(int, (F) || (k) <= GK_NCAP, sh.rowptr[k], a.rowptr[nb + (k)])) - e0)
  // (LDS read from a clamped slot, overridden under a uniform branch when the node is outside the table: the two
  //  loads must not be merged into one pointer select)
#define A_NM(F, jl)                                                                            \
  ({                                                                                           \
    const bool in_ = (unsigned)(jl) < (unsigned)min(n, GK_NCAP);                               \
    float v_ = sh.nm[in_ ? (jl) : 0];                                                          \
    if (!in_) v_ = a.node_mask[nb + (jl)];                                                     \
    unif(v_);                                                                                  \
  })
  // logit table: LDS for the first GK_ECAP slots, the alpha output rows (overwritten later) beyond; per lane group
#define A_LG(F, t, e) A_SEL(float, (F) || (t) < GK_ECAP, sh.lg[(t) * HS + grp], a.alpha[(size_t)(e) * a.H + hd])
  // x_l row jl of this graph, float4 column c of the slice: LDS when staged, global otherwise
#define A_XL(jl, c)                                                                            \
  ({                                                                                           \
    const bool in_ = (unsigned)(jl) < (unsigned)rows;                                          \
    float4 v_ = s_xl[(in_ ? (jl) : 0) * RQ + (c)];                                             \
    if (!in_) v_ = a.x_l[(size_t)(nb + (jl)) * R + hoff + (c)];                                \
    v_;                                                                                        \
  })

  // ---- phase B: edge-parallel logits.  UU e_proj rows (streamed) + UU x_r rows requested together per wave ------
#define GK_PHASE_B(F, UU)                                                                                      \
  _Pragma("unroll 1") for (int tb = wave * (UU); tb < ne; tb += GK_WAVES * (UU)) {                             \
    float4 epv[UU][P], xrv[UU][P];                                                                             \
    _Pragma("unroll") for (int u = 0; u < (UU); ++u) {                                                         \
      const int t = tb + u;                                                                                    \
      if (t < ne) {                                                                                            \
        const float4 *ep = a.e_proj + (size_t)A_EID(F, t) * R + hoff;                                          \
        const float4 *xr = a.x_r + (size_t)(nb + A_DST(F, t)) * R + hoff;                                      \
        _Pragma("unroll") for (int p = 0; p < P; ++p) {                                                        \
          epv[u][p] = ok[p] ? ld_stream(ep + off[p], nt) : make_float4(0.f, 0.f, 0.f, 0.f);                    \
          xrv[u][p] = ok[p] ? xr[off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);                                    \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int u = 0; u < (UU); ++u) {                                                         \
      const int t = tb + u;                                                                                    \
      if (t < ne) {                                                                                            \
        const int jl = A_SRC(F, t) - nb;                                                                       \
        float me = 1.f;                                                                                        \
        if (mode == 1) me = A_NM(F, jl) * A_NM(F, A_DST(F, t));                                                \
        else if (mode == 2) me = unif(a.edge_mask[A_EID(F, t)]);                                               \
        float part = 0.f;                                                                                      \
        _Pragma("unroll") for (int p = 0; p < P; ++p) {                                                        \
          if (ok[p]) {                                                                                         \
            const float4 v = epv[u][p], r4 = xrv[u][p];                                                        \
            const float4 w4 = A_XL(jl, off[p]);                                                                \
            float4 s;                                                                                          \
            s.x = (r4.x + w4.x) + v.x;                                                                         \
            s.y = (r4.y + w4.y) + v.y;                                                                         \
            s.z = (r4.z + w4.z) + v.z;                                                                         \
            s.w = (r4.w + w4.w) + v.w;                                                                         \
            if (mode != 0) { s.x *= me; s.y *= me; s.z *= me; s.w *= me; }                                     \
            s.x = leaky(s.x, slope); s.y = leaky(s.y, slope); s.z = leaky(s.z, slope); s.w = leaky(s.w, slope); \
            if (mode != 0) { s.x *= me; s.y *= me; s.z *= me; s.w *= me; }                                     \
            part += dot4(s, att4[p]);                                                                          \
          }                                                                                                    \
        }                                                                                                      \
        const float logit = group_sum<G>(part);                                                                \
        if (l == 0) {                                                                                          \
          if ((F) || t < GK_ECAP) sh.lg[t * HS + grp] = logit;                                                 \
          else a.alpha[(size_t)A_EID(F, t) * a.H + hd] = logit;                                                \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
  }

  // ---- phase C: node-parallel softmax + aggregation; operands in LDS ---------------------------------------------
#define GK_PHASE_C(F)                                                                                          \
  _Pragma("unroll 1") for (int k = wave; k < n; k += GK_WAVES) {                                               \
    const int i = nb + k;                                                                                      \
    const int rb = A_RP(F, k), re = A_RP(F, k + 1);                                                            \
    float mx = -INFINITY;                                                                                      \
    _Pragma("unroll 1") for (int t = rb; t < re; ++t) mx = fmaxf(mx, A_LG(F, t, A_EID(F, t)));                 \
    float den = 0.f;                                                                                           \
    _Pragma("unroll 1") for (int t = rb; t < re; ++t) den += expf(A_LG(F, t, A_EID(F, t)) - mx);               \
    den += 1e-16f;                                                                                             \
    const float mi = mode == 1 ? A_NM(F, k) : 1.f;                                                             \
    float4 acc[P];                                                                                             \
    _Pragma("unroll") for (int p = 0; p < P; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);                    \
    _Pragma("unroll 1") for (int t = rb; t < re; ++t) {                                                        \
      const int jl = A_SRC(F, t) - nb, e = A_EID(F, t);                                                        \
      const float w = expf(A_LG(F, t, e) - mx) / den;                                                          \
      if (l == 0) a.alpha[(size_t)e * a.H + hd] = w;                                                           \
      float wm = w;                                                                                            \
      if (mode == 1) wm = __fmul_rn(w, A_NM(F, jl) * mi);                                                      \
      else if (mode == 2) wm = __fmul_rn(w, unif(a.edge_mask[e]));                                             \
      _Pragma("unroll") for (int p = 0; p < P; ++p) {                                                          \
        if (ok[p]) {                                                                                           \
          const float4 u4 = A_XL(jl, off[p]);                                                                  \
          acc[p].x = __fadd_rn(acc[p].x, __fmul_rn(u4.x, wm));                                                 \
          acc[p].y = __fadd_rn(acc[p].y, __fmul_rn(u4.y, wm));                                                 \
          acc[p].z = __fadd_rn(acc[p].z, __fmul_rn(u4.z, wm));                                                 \
          acc[p].w = __fadd_rn(acc[p].w, __fmul_rn(u4.w, wm));                                                 \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int p = 0; p < P; ++p) {                                                            \
      if (ok[p]) {                                                                                             \
        float4 o = acc[p];                                                                                     \
        if (a.bias) {                                                                                          \
          const float4 b = a.bias[hoff + off[p]];                                                              \
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;                                                      \
        }                                                                                                      \
        st_stream(a.out + (size_t)i * R + hoff + off[p], o, nt);                                               \
      }                                                                                                        \
    }                                                                                                          \
  }

  // flags bits 3/4 are ablation switches for profiling (skip a phase; outputs are then wrong)
  if (!(a.flags & 8)) {
    if (fast) { GK_PHASE_B(true, U) } else { GK_PHASE_B(false, 1) }
  }
  __syncthreads();   // logits (LDS and, for huge graphs, global) visible to the whole workgroup
  if (!(a.flags & 16)) {
    if (fast) { GK_PHASE_C(true) } else { GK_PHASE_C(false) }
  }

#undef A_SEL
#undef A_XL
#undef A_SRC
#undef A_EID
#undef A_DST
#undef A_RP
#undef A_NM
#undef A_LG
#undef GK_PHASE_B
#undef GK_PHASE_C
}

template <int HS, int P>
static int launch_one(const MpArgs &a, int lds_budget, int nmax_host, hipStream_t st) {
  constexpr int U = P == 1 ? 4 : 2;
  const size_t row_bytes = (size_t)HS * a.C * 4;
  const size_t static_bytes = sizeof(GkShared<HS>);
  if ((size_t)lds_budget < static_bytes + 4 * row_bytes) return ISG_EUNSUPPORTED;
  MpArgs b = a;
  b.lrows = (int)(((size_t)lds_budget - static_bytes) / row_bytes);
  if (b.lrows > nmax_host) b.lrows = nmax_host;   // no point in reserving more rows than the largest graph has
  const size_t dyn = (size_t)b.lrows * row_bytes;
  const long long items = (long long)a.B * (a.H / HS);
  if (items >= (1ll << 31)) return ISG_EUNSUPPORTED;
  dim3 grid((unsigned)items), block(GK_THREADS);
  gatv2_mp_graph_kernel<HS, P, U><<<grid, block, dyn, st>>>(b);
  return check_launch();
}

int launch_mp_graph(MpArgs a, int nmax_host, hipStream_t st) {
  const int Q = a.C >> 2;
  // heads per workgroup: the largest HS | H with a row slice of at most 1280 bytes (at least one head)
  int HS = 1;
  for (int hs = 8; hs >= 1; hs >>= 1)
    if (a.H % hs == 0 && hs * a.C * 4 <= 1280) { HS = hs; break; }
  if (const char *f = getenv("ISG_MP_HS")) {   // experiment override
    const int hs = atoi(f);
    if (hs > 0 && a.H % hs == 0 && 64 % hs == 0) HS = hs;
  }
  const int G = 64 / HS;
  const int P = (Q + G - 1) / G;
  // static + dynamic LDS per workgroup: 40 KB -> 4 workgroups (32 waves) per CU of 160 KB
  const char *kb = getenv("ISG_MP_LDS_KB");
  const int budget = (kb ? atoi(kb) : 40) * 1024;
#define ISG_GK(hs, p) if (HS == hs && P == p) return launch_one<hs, p>(a, budget, nmax_host, st)
  ISG_GK(1, 1); ISG_GK(1, 2); ISG_GK(1, 3); ISG_GK(1, 4);
  ISG_GK(2, 1); ISG_GK(2, 2);
  ISG_GK(4, 1); ISG_GK(4, 2);
  ISG_GK(8, 1); ISG_GK(8, 2);
#undef ISG_GK
  return ISG_EUNSUPPORTED;
}

}  // namespace isg
