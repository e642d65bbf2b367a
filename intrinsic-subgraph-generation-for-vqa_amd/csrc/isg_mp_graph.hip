// GATv2 message passing, per-graph form: the x_l rows of one scene graph live in LDS.
//
// Why: in the node-chunk kernel (isg_mp.hip) every edge gathers its source row x_l[j] from L2/HBM; PMC
// counters showed ~1.5x the algorithmic read traffic (profiles/r01_b_mp_traffic.md).  A PyG batch keeps a
// graph's nodes contiguous and its edges inside the graph, so one workgroup can stage the graph's x_l slice
// with coalesced loads ONCE and serve every x_l[j] (logit pass and aggregation pass) from LDS.  HBM then
// sees x_l, x_r, out exactly once per row and e_proj once per edge: the algorithmic minimum.
//
//   grid  = (graphs, H / HS): a workgroup owns one graph and HS consecutive heads (HS*C*4 bytes <= ~1.25 KB per
//           row keeps a 36-node graph at <= 36-45 KB of LDS, i.e. 3-4 workgroups per CU)
//   block = 8 waves; wave w owns destination nodes w, w+8, ... of the graph
//   lanes = HS groups of G = 64/HS lanes (group = head), lane l owns float4 columns l, l+G, ... (P passes)
//   LDS   = x_l slice rows [lrows][HS*C] (dynamic) + the graph's CSR (rowptr, src, eid) + a logit strip per wave
// A source outside the staged window (graph larger than lrows rows, or an edge that leaves its graph) is read
// from global memory, so correctness never depends on the batch layout.
// Arithmetic, order of operations and roundings are those of the node-chunk kernel (see isg_mp.hip).
#include "isg_mp.hpp"

namespace isg {

constexpr int GK_WAVES = 8;
constexpr int GK_NCAP = 128;   // rowptr entries of a graph staged in LDS
constexpr int GK_ECAP = 512;   // CSR slots of a graph staged in LDS
constexpr int GK_U = 4;        // edges whose e_proj rows a wave requests together

template <int HS, int P>
__global__ __launch_bounds__(GK_WAVES * 64) void gatv2_mp_graph_kernel(MpArgs a) {
  constexpr int G = 64 / HS;
  extern __shared__ __attribute__((aligned(16))) float4 s_xl[];   // [lrows][HS*Q]
  __shared__ int s_rowptr[GK_NCAP + 4];           // sizes kept multiples of 16 B: the dynamic base stays aligned
  __shared__ int s_src[GK_ECAP];
  __shared__ int s_eid[GK_ECAP];
  __shared__ float s_logit[GK_WAVES][MP_LCAP * HS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, hg = blockIdx.y;
  const int nb = a.graph_ptr[g];
  const int n = a.graph_ptr[g + 1] - nb;
  if (n <= 0) return;
  const int Q = a.C >> 2;        // float4 per head
  const int R = a.H * Q;         // float4 per full row
  const int RQ = HS * Q;         // float4 per staged row slice
  const int hoff = hg * RQ;      // first float4 of this workgroup's head slice inside a row
  const int rows = min(n, a.lrows);
  const int e0 = a.rowptr[nb], e1 = a.rowptr[nb + n];
  const int ne = min(e1 - e0, GK_ECAP);

  for (int t = tid; t <= min(n, GK_NCAP); t += GK_WAVES * 64) s_rowptr[t] = a.rowptr[nb + t];
  for (int t = tid; t < ne; t += GK_WAVES * 64) {
    s_src[t] = a.src[e0 + t];
    s_eid[t] = a.eid[e0 + t];
  }
  for (int idx = tid; idx < rows * RQ; idx += GK_WAVES * 64) {
    const int r = idx / RQ, c = idx - r * RQ;
    s_xl[idx] = a.x_l[(size_t)(nb + r) * R + hoff + c];
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
  const int mode = a.edge_mask ? 2 : (a.node_mask ? 1 : 0);
  const int hd = hg * HS + grp;   // global head index of this lane group
  float *s_lg = s_logit[wave];

  for (int k = wave; k < n; k += GK_WAVES) {
    const int i = nb + k;
    const int rb = k < GK_NCAP ? s_rowptr[k] : a.rowptr[i];
    const int re = k + 1 <= GK_NCAP ? s_rowptr[k + 1] : a.rowptr[i + 1];
    float4 xr4[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
      xr4[p] = ok[p] ? a.x_r[(size_t)i * R + hoff + off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float mi = mode == 1 ? a.node_mask[i] : 1.f;

    // ---- pass 1: logits + per-head maximum; e_proj rows of GK_U edges are requested together so a wave keeps
    //      GK_U KB of HBM loads in flight instead of one dependent load per edge ----------------------------------
    float mx = -INFINITY;
    for (int t0 = rb; t0 < re; t0 += GK_U) {
      int jj[GK_U], ee[GK_U];
      float mm[GK_U];
      float4 epv[GK_U][P];
#pragma unroll
      for (int u = 0; u < GK_U; ++u) {
        const int t = t0 + u;
        jj[u] = 0; ee[u] = 0; mm[u] = 1.f;
        if (t < re) {
          const int rel = t - e0;
          if (rel < GK_ECAP) { jj[u] = s_src[rel]; ee[u] = s_eid[rel]; } else { jj[u] = a.src[t]; ee[u] = a.eid[t]; }
          const float4 *ep = a.e_proj + (size_t)ee[u] * R + hoff;
#pragma unroll
          for (int p = 0; p < P; ++p) epv[u][p] = ok[p] ? ep[off[p]] : make_float4(0.f, 0.f, 0.f, 0.f);
          if (mode == 1) mm[u] = a.node_mask[jj[u]] * mi;
          else if (mode == 2) mm[u] = a.edge_mask[ee[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < GK_U; ++u) {
        const int t = t0 + u;
        if (t < re) {
          const int j = jj[u], e = ee[u];
          const float me = mm[u];
          const unsigned jl = (unsigned)(j - nb);
          const bool in_lds = jl < (unsigned)rows;
          const float4 *xl_g = a.x_l + (size_t)j * R + hoff;
          const float4 *xl_s = s_xl + (size_t)(in_lds ? jl : 0) * RQ;
          float part = 0.f;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if (ok[p]) {
              const float4 v = epv[u][p];
              const float4 w4 = in_lds ? xl_s[off[p]] : xl_g[off[p]];
              float4 s;
              s.x = (xr4[p].x + w4.x) + v.x;
              s.y = (xr4[p].y + w4.y) + v.y;
              s.z = (xr4[p].z + w4.z) + v.z;
              s.w = (xr4[p].w + w4.w) + v.w;
              if (mode != 0) { s.x *= me; s.y *= me; s.z *= me; s.w *= me; }
              s.x = leaky(s.x, a.slope); s.y = leaky(s.y, a.slope); s.z = leaky(s.z, a.slope); s.w = leaky(s.w, a.slope);
              if (mode != 0) { s.x *= me; s.y *= me; s.z *= me; s.w *= me; }
              part += dot4(s, att4[p]);
            }
          }
          const float logit = group_sum<G>(part);
          mx = fmaxf(mx, logit);
          const int slot = t - rb;
          if (l == 0) {
            if (slot < MP_LCAP) s_lg[slot * HS + grp] = logit;
            else a.alpha[(size_t)e * a.H + hd] = logit;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- pass 2: denominator in edge order ----------------------------------------------------------------
    float den = 0.f;
    for (int t = rb; t < re; ++t) {
      const int slot = t - rb;
      float lg;
      if (slot < MP_LCAP) {
        lg = s_lg[slot * HS + grp];
      } else {
        const int rel = t - e0;
        const int e = rel < GK_ECAP ? s_eid[rel] : a.eid[t];
        lg = l == 0 ? a.alpha[(size_t)e * a.H + hd] : 0.f;
        lg = __shfl(lg, lane - l, 64);
      }
      den += expf(lg - mx);
    }
    den += 1e-16f;

    // ---- pass 3: alpha out + aggregation from the LDS-resident rows -----------------------------------------
    float4 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = rb; t < re; ++t) {
      const int rel = t - e0;
      int j, e;
      if (rel < GK_ECAP) { j = s_src[rel]; e = s_eid[rel]; } else { j = a.src[t]; e = a.eid[t]; }
      const int slot = t - rb;
      float lg;
      if (slot < MP_LCAP) {
        lg = s_lg[slot * HS + grp];
      } else {
        lg = l == 0 ? a.alpha[(size_t)e * a.H + hd] : 0.f;
        lg = __shfl(lg, lane - l, 64);
      }
      const float w = expf(lg - mx) / den;
      if (l == 0) a.alpha[(size_t)e * a.H + hd] = w;
      float wm = w;
      if (mode == 1) wm = __fmul_rn(w, a.node_mask[j] * mi);
      else if (mode == 2) wm = __fmul_rn(w, a.edge_mask[e]);
      const unsigned jl = (unsigned)(j - nb);
      const bool in_lds = jl < (unsigned)rows;
      const float4 *xl_g = a.x_l + (size_t)j * R + hoff;
      const float4 *xl_s = s_xl + (size_t)(in_lds ? jl : 0) * RQ;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (ok[p]) {
          const float4 u = in_lds ? xl_s[off[p]] : xl_g[off[p]];
          acc[p].x = __fadd_rn(acc[p].x, __fmul_rn(u.x, wm));
          acc[p].y = __fadd_rn(acc[p].y, __fmul_rn(u.y, wm));
          acc[p].z = __fadd_rn(acc[p].z, __fmul_rn(u.z, wm));
          acc[p].w = __fadd_rn(acc[p].w, __fmul_rn(u.w, wm));
        }
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (ok[p]) {
        float4 o = acc[p];
        if (a.bias) {
          const float4 b = a.bias[hoff + off[p]];
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        a.out[(size_t)i * R + hoff + off[p]] = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int HS, int P>
static int launch_one(const MpArgs &a, size_t dyn_bytes, hipStream_t st) {
  dim3 grid((unsigned)a.B, (unsigned)(a.H / HS)), block(GK_WAVES * 64);
  gatv2_mp_graph_kernel<HS, P><<<grid, block, dyn_bytes, st>>>(a);
  return check_launch();
}

int launch_mp_graph(MpArgs a, int nmax_host, hipStream_t st) {
  const int Q = a.C >> 2;
  // heads per workgroup: the largest HS | H with a row slice of at most 1280 bytes (at least one head)
  int HS = 1;
  for (int hs = 8; hs >= 1; hs >>= 1)
    if (a.H % hs == 0 && hs * a.C * 4 <= 1280) { HS = hs; break; }
  const int G = 64 / HS;
  const int P = (Q + G - 1) / G;
  const size_t row_bytes = (size_t)HS * a.C * 4;
  const size_t budget = 40 * 1024;    // static + dynamic per workgroup -> 4 workgroups per CU (160 KB LDS)
  const size_t static_bytes = (GK_NCAP + 4) * 4 + 2 * GK_ECAP * 4 + (size_t)GK_WAVES * MP_LCAP * HS * 4;
  int lrows = (int)((budget - static_bytes) / row_bytes);
  if (lrows < 4) return ISG_EUNSUPPORTED;
  if (lrows > nmax_host) lrows = nmax_host;
  a.lrows = lrows;
  const size_t dyn = (size_t)lrows * row_bytes;
#define ISG_GK(hs, p) if (HS == hs && P == p) return launch_one<hs, p>(a, dyn, st)
  ISG_GK(1, 1); ISG_GK(1, 2); ISG_GK(1, 3); ISG_GK(1, 4);
  ISG_GK(2, 1); ISG_GK(2, 2);
  ISG_GK(4, 1); ISG_GK(4, 2);
  ISG_GK(8, 1); ISG_GK(8, 2);
#undef ISG_GK
  return ISG_EUNSUPPORTED;
}

}  // namespace isg
