// GATv2 attention logits WITHOUT the projected edge features in memory.
//
// MaskingGATv2Conv (ISubGVQA/models/mgat_v2_conv.py:215-279) scores an edge e = (j -> i) per head h as
//     logit[e, h] = att_h . leaky_relu(x_l[j] + x_r[i] + lin_edge(edge_attr[e]))_h
// and uses e_proj = lin_edge(edge_attr) for nothing else: the message is alpha * x_l[j].  The un-fused pair writes e_proj
// (E x H*C fp32: 420 MB per layer at BASELINE configs[1]) in isg_linear_f16x3 and streams it back in the message-passing
// kernel -- the largest HBM stream of the step, for a tensor that only ever contributes E x H scalars.  Here the panel
// GEMM of isg_gemm_f16x3.hip keeps its result in the accumulators and finishes the logit in its epilogue; the
// message-passing kernel (isg_gatv2_mp_fwd_logits) then takes logits[E, H] (3 MB) instead of e_proj and x_r.
//
//   order      CSR SLOT order (edges sorted by destination, a graph's slots contiguous): consecutive slots share x_r[i]
//              and draw x_l[j] from one graph's ~20 rows, so the row gathers hit L1 / L2; the logits come out in the order
//              the message-passing kernel stages them
//   workgroup  64 slots, 8 waves; the 64 edge_attr rows (gathered by edge id, 512 B each) are scaled per row, split into
//              (hi, mid) fp16 planes and staged in LDS once (isg_linear_f16x3's A-stationary panel)
//   wave       one 32-slot half of the panel and every fourth 32-channel tile of the row; a tile's product is formed
//              TRANSPOSED -- the W fragment is the MFMA's A operand, the edge panel its B operand -- so a lane holds ONE edge
//              (lane & 31) and 16 channels of it in its accumulator registers: the sum over channels is in-lane, and the
//              two half-waves (channels +4) meet in one cross-lane add per head.  (With the panel as the A operand a lane
//              holds one channel of 16 edges and every (edge, tile) needs a 32-lane reduction: that form was measured in
//              profiles/r02_d_fused_edge.md and lost.)
//   epilogue   per tile: e = acc * s_row * s_col (exact powers of two), z = (x_r[i] + x_l[j]) + e, mask, leaky, mask,
//              z * att accumulated per lane; the 2 x 4 x 16 B row gathers of a tile are issued before its MFMA loop
// W never touches LDS (fragment-major planes from L2, one k-step ahead), exactly as in isg_linear_f16x3.
//
// Round 5: that is the PANEL kernel, which now serves K < 128 only.  K = 128 and 128 < K <= 304 run the ROWS kernel further down
// (gatv2_edge_logits_rows_kernel): the roles swapped -- edge rows resident in registers, W tiles through a three-slot LDS ring
// requested by a wave of their own, once per 224 slots -- which is the "W tile kept in LDS across several panels" the notes
// below ask for: 180 us against 228 at K = 128, 520 against 1 110 at the reference's own C = 300 / K = 300.
#include "isg_f16x3.hpp"
#include "isg_mp.hpp"
#include "isg_diag.hpp"

#include <stdlib.h>
#include <type_traits>


namespace isg {

constexpr int EL_BM = 64, EL_KC = 128, EL_THREADS = 512;

struct ElArgs {
  const float *edge_attr;           // rows by EDGE ID, stride lda
  const _Float16 *Wf;               // fragment-major (hi, mid) planes of lin_edge.weight [H*C, K] (isg_split_f16x2_frag)
  const float *w_inv;               // [NT * 32]
  const float *x_l, *x_r;           // rows by node id, strides ldl / ldr (column slices of the fused lin_l | lin_r output)
  const float *att;                 // [H * C]
  const int *eid, *src, *dst;       // CSR slot order
  const float *edge_mask, *node_mask;   // optional (NULL: the layer is not masked)
  float *logits;                    // [E, H], slot order
  int E, H, C, K, KS, NT, lda, ldl, ldr;
  int Cp, LD;                       // channels per head padded to whole 32-channel tiles (== C when 32 | C); panel row pitch in halfs
  int64_t hsl, hsr;                 // distance between a row's consecutive HEAD slices in x_l / x_r (floats): C when the heads
                                    // lie side by side in one row (row-major [N, H*C]); N * C for a head-major [H][N][C] tensor
  float slope;
};

// What was measured on the way to this form, all at BASELINE configs[1] on one MI355X, same-box A/B (first version: 4 waves,
// a wave = 64 slots x one head, 194 VGPRs, 2 waves per SIMD: 214 us; profiles/r02_w_edge_logits.md has the tables):
//   * ablation of the first version: 112 us without the x_l / x_r gathers, 180 without the MFMAs, 171 without the W loads
//   * v_mfma_f32_16x16x32_f16 layout (a lane's channels contiguous, half the rows per gather instruction): 218 us
//   * gathers issued a whole tile ahead (second register set): 226 us
//   * the four waves on neighbouring 128-byte pieces of the same rows at the same time: 216 us
//   * all W fragments of a tile ahead of its gathers (so that an in-order vmcnt wait for a fragment does not drain the
//     gathers): 64 + 64 + 32 registers plus the epilogue's temporaries do not fit 256 (56-96 spilled); not run
//   * this form (a wave = ONE 32-slot half, 8 waves, 97 VGPRs, four waves per SIMD): 217 us; its ablation: 122 us without
//     the gathers, 232 without the MFMAs (hidden), 118 without both
//   * x_l / x_r as column halves of [N, 2HC] (4 KB row pitch), as two dense tensors (2 KB), head-major [2H][N][C] (512-byte
//     rows, 8 per page): 228 / 228 / 229 us, identical logits
//   * counters: HBM fetch 508 MB per launch = 1.15x the algorithmic 442 MB; L2 hit rate 67 %; 71 M L1 line accesses (the
//     plain lin_edge GEMM: 23 M); TCP_UTCL1_STALL_INFLIGHT_MAX 24.6 M (10.0 M)
//   * in-kernel stamps (a diagnostic build of round 2, gone from the source with its ablation arms: the tables are
//     profiles/r02_w_edge_logits.md): a wave lives 68k cycles for its 4 tiles: staging 18 %,
//     issuing the gathers 10 % (1.8k cycles per tile: back-pressure), k loops 57 % (9.8k cycles per tile for 768 cycles of
//     MFMA), residual gather wait after the k loop 0.1 %, epilogue 8 %: the gathers' latency under load (3-4 us per tile) is
//     paid at the first W-fragment wait of every k loop -- vector-memory operations retire in order
//   * a 12-wave workgroup with 4 loader waves (rows as coalesced 512-byte head slices, pre-added into LDS) and 8 matrix waves
//     that never gather -- the form that removes exactly that serialisation: 223 us against 216 us for this kernel
// So: ten forms within 211-232 us.  The stamps show where a wave waits, but a kernel without that wait takes the same time:
// the limit is chip-wide (per launch ~0.5 GB from HBM, 0.8-1.6 GB of W fragments and ~0.84 GB of row requests through L2);
// the next attempt should shrink that traffic (a W tile kept in LDS across several panels), not reorder it.
// The pair (this kernel + isg_gatv2_mp_fwd_logits) is 313-320 us against 334-350 us for isg_linear_f16x3 +
// isg_gatv2_mp_fwd, and the configs[1] step 2.20-2.22 ms against 2.24-2.29 ms on the same box.
// (A form that also computed x_r = lin_r(x) here, from a second panel of x[dst] rows, was built in round 2 and measured 2.255 vs
// 2.235 ms per step: removed in round 4.)
// PASSES: 128-column passes of the panel staging (1: K <= 128, the only instantiation: wider edge rows take the rows kernel below).
// Head dimensions that are not a multiple of 32 (the reference's C = 300) run on heads PADDED to Cp = 320 channels: the caller's W
// fragments are those of the weight with zero rows behind every head's C-th (isg_split_f16x2_frag of the padded matrix), att is
// staged with zeros there, and the row gathers skip the float4 pieces beyond a head's last channel (C % 4 == 0: a piece is whole
// or absent) -- the padding contributes exact zeros to the logit.
template <bool MASKED, int PASSES>
__global__ __launch_bounds__(EL_THREADS, 4) void gatv2_edge_logits_kernel(ElArgs a) {
  __shared__ float s_inv[EL_BM];
  extern __shared__ __attribute__((aligned(16))) unsigned char el_smem[];     // panel [2][64][LD] halfs, then att / w_inv / partials
  const int LD = a.LD;
  _Float16 *sA = reinterpret_cast<_Float16 *>(el_smem);
#define EL_SA(q, row, k) (sA + ((q) * EL_BM + (row)) * LD + (k))
  float *s_cw = reinterpret_cast<float *>(el_smem + (size_t)2 * EL_BM * LD * 2);     // att [H*Cp], w_inv [H*Cp], partial logits [4][64][H]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, tw = wave & 3;     // which 32 slots of the panel; which channel tiles (tw, tw + 4, ...)
  const int m0 = blockIdx.x * EL_BM;
  const int fr = lane & 31, hh = lane >> 5, fk = hh * 8;
  const int HC = a.H * a.Cp;                      // PADDED width: the index space of att / w_inv in LDS and of the channel tiles

  // ---- stage the edge panel once: rows gathered by edge id -> row scale -> (hi, mid) planes -------------------------------
  {
    const int KP4 = a.KS * 4;                     // float4 columns of the padded panel (zeros beyond K)
    float4 ra[4][PASSES];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + EL_THREADS * u;
      const int row = i >> 5, c4 = i & 31;
      const int e = a.eid[min(m0 + row, a.E - 1)];
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int gk = min((c4 + 32 * ps) * 4, a.K - 4);
        ra[u][ps] = *reinterpret_cast<const float4 *>(a.edge_attr + (int64_t)e * a.lda + gk);
      }
    }
    for (int c = tid; c < HC; c += EL_THREADS) {
      const int hd = c / a.Cp, ch = c - hd * a.Cp;
      s_cw[c] = ch < a.C ? a.att[hd * a.C + ch] : 0.f;
      s_cw[HC + c] = a.w_inv[c];
    }
    for (int c = tid; c < 4 * EL_BM * a.H; c += EL_THREADS) s_cw[2 * HC + c] = 0.f;      // the waves' partial logits
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + EL_THREADS * u;
      const int row = i >> 5, c4 = i & 31;       // the 32 lanes of a half-wave hold one row
      float mx = 0.f;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        float4 &v = ra[u][ps];
        if (m0 + row >= a.E || (c4 + 32 * ps) * 4 >= a.K) v = make_float4(0.f, 0.f, 0.f, 0.f);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
      mx = group_max<32>(mx);
      float s, inv;
      h3_scale(mx, s, inv);
      if (c4 == 0) s_inv[row] = inv;
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        if (c4 + 32 * ps >= KP4) continue;
        float4 v = ra[u][ps];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        hf16x4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        hf16x4 mid = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                      (_Float16)(v.w - (float)hi[3])};
        *reinterpret_cast<hf16x4 *>(EL_SA(0, row, (c4 + 32 * ps) * 4)) = hi;
        *reinterpret_cast<hf16x4 *>(EL_SA(1, row, (c4 + 32 * ps) * 4)) = mid;
      }
    }
  }


  // ---- this lane's edge (slot m0 + half * 32 + fr): endpoints, mask value ------------------------------------------------
  const int prow = half * 32 + fr;               // row of the panel
  const int sl = min(m0 + prow, a.E - 1);
  const int s_node = a.src[sl], d_node = a.dst[sl];
  const int64_t xl_off = (int64_t)s_node * a.ldl + 4 * hh, xr_off = (int64_t)d_node * a.ldr + 4 * hh;
  float me = 1.f;
  if (MASKED) me = a.edge_mask ? a.edge_mask[a.eid[sl]] : a.node_mask[s_node] * a.node_mask[d_node];
  __syncthreads();
  const float sinv = s_inv[prow];
  const int KS = a.KS;
  const unsigned plane_b = (unsigned)a.NT * (unsigned)KS * 1024u;      // bytes per W plane
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.Wf), 0, (int)(2u * plane_b), 0x00020000);
  const int voff = lane * 16;
  const int tph = a.Cp >> 5;           // channel tiles per (padded) head
  const float slope = a.slope;

  // In round r the four tile-waves of a half take channel tiles 4 r .. 4 r + 3.  A wave's partial logits (its tiles of a
  // head) go to its own LDS slots and are summed over the tile-waves in a fixed order at the end.
  float *s_part = s_cw + 2 * HC;          // [4 tile-waves][64 slots][H]
  float part[4];             // four partial sums (one per channel group g), added pairwise: short chains
  int cur_hd = -1;
  auto flush = [&](int hd) {
    const float mine = (part[0] + part[1]) + (part[2] + part[3]);
    const float tot = mine + __shfl_xor(mine, 32);
    if (hh == 0) s_part[(tw * EL_BM + prow) * a.H + hd] = tot;
  };
#pragma unroll 1
  for (int nt = tw; nt < a.NT; nt += 4) {
    const int hd = nt / tph;
    if (hd != cur_hd) {
      if (cur_hd >= 0) flush(cur_hd);
      cur_hd = hd;
#pragma unroll
      for (int g = 0; g < 4; ++g) part[g] = 0.f;
    }
    const int cb = nt * 32 + 4 * hh;         // this lane's channels of the tile: cb + 8 * g + j, g = r >> 2, j = r & 3
    const int64_t col_l = (int64_t)hd * a.hsl + (nt - hd * tph) * 32, col_r = (int64_t)hd * a.hsr + (nt - hd * tph) * 32;
    float4 xl[4], xr[4];
    const int cin = (nt - hd * tph) * 32 + 4 * hh;          // this lane's first channel of the tile inside its head
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (cin + 8 * g < a.C) {                               // (always, when 32 | C)
        xl[g] = *reinterpret_cast<const float4 *>(a.x_l + xl_off + col_l + 8 * g);
        xr[g] = *reinterpret_cast<const float4 *>(a.x_r + xr_off + col_r + 8 * g);
      } else {
        xl[g] = xr[g] = make_float4(0.f, 0.f, 0.f, 0.f);     // a padded channel: W rows and att are zero there
      }
    }
    hf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned wb = (unsigned)nt * (unsigned)KS * 1024u;
    hf16x8 a0[2], a1[2], w0[2], w1[2];
#define EL_LOAD_W(W, s)                                                                                          \
  _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                  \
      W[q] = __builtin_bit_cast(hf16x8, __builtin_amdgcn_raw_buffer_load_b128(                                   \
          wrsrc, voff, (int)(wb + q * plane_b + (unsigned)(s) * 1024u), 0));
#define EL_LOAD_A(Afr, ksl)                                                                                      \
  _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                  \
      Afr[q] = *reinterpret_cast<const hf16x8 *>(EL_SA(q, prow, (ksl) * 16 + fk));
    // transposed product: W fragment = A operand (rows = channels), edge panel = B operand (columns = edges)
#define EL_MMA(Afr, W)                                                                                           \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[0], Afr[1], acc, 0, 0, 0);                                      \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[1], Afr[0], acc, 0, 0, 0);                                      \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[0], Afr[0], acc, 0, 0, 0);
    EL_LOAD_W(w0, 0)
    EL_LOAD_A(a0, 0)
    int ks = 0;
#pragma unroll 1
    for (; ks + 2 <= KS; ks += 2) {
      EL_LOAD_W(w1, ks + 1)
      EL_LOAD_A(a1, ks + 1)
      EL_MMA(a0, w0)
      EL_LOAD_W(w0, min(ks + 2, KS - 1))
      EL_LOAD_A(a0, min(ks + 2, KS - 1))
      EL_MMA(a1, w1)
    }
    if (ks < KS) { EL_MMA(a0, w0) }
#undef EL_LOAD_W
#undef EL_LOAD_A
#undef EL_MMA
    // ---- epilogue of the tile: the logit's partial sums over this lane's 16 channels ---------------------------------------
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 at4 = *reinterpret_cast<const float4 *>(&s_cw[cb + 8 * g]);
      const float4 wi4 = *reinterpret_cast<const float4 *>(&s_cw[HC + cb + 8 * g]);
      const float atv[4] = {at4.x, at4.y, at4.z, at4.w}, wiv[4] = {wi4.x, wi4.y, wi4.z, wi4.w};
      const float lv[4] = {xl[g].x, xl[g].y, xl[g].z, xl[g].w};
      const float rv[4] = {xr[g].x, xr[g].y, xr[g].z, xr[g].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = (acc[g * 4 + j] * sinv) * wiv[j];     // both scales are powers of two: exact
        float z = (rv[j] + lv[j]) + e;
        if (MASKED) z *= me;
        z = leaky(z, slope);
        if (MASKED) z *= me;
        part[g] = fmaf(z, atv[j], part[g]);
      }
    }
  }
  if (cur_hd >= 0) flush(cur_hd);
  __syncthreads();
  for (int c = tid; c < EL_BM * a.H; c += EL_THREADS) {       // (slot, head): the tile-waves' partials in a fixed order
    const float v = (s_part[c] + s_part[EL_BM * a.H + c]) + (s_part[2 * EL_BM * a.H + c] + s_part[3 * EL_BM * a.H + c]);
    const int slot = m0 + c / a.H;
    if (slot < a.E) a.logits[(int64_t)slot * a.H + (c % a.H)] = v;
  }
#undef EL_SA
}

// ---------------------------------------------------------------------------------------------------------------------
// The same logits for WIDE layers (the reference's own: C = 300, 300 edge features).  The panel kernel above streams all of
// lin_edge's fragments from L2 once per 64 slots; at H Cp = 1280, K = 304 that is 1.56 MB per workgroup, 5 GB per launch, and with an
// 80 KB panel only one workgroup fits a CU: 0.9-1.1 ms per launch against 0.6 ms for lin_edge as a plain GEMM (profiles/
// r05_u_edge_logits_wide.txt).  Here the roles are swapped: a wave keeps ITS 32 slots' edge rows -- scaled and split -- in
// registers for the whole kernel (K padded to 16 KS_T: 8 KS_T registers), and the weight tiles stream through LDS, each fetched
// ONCE per workgroup by LDS-DMA into a THREE-slot ring and read by all the computing waves.  A wave owns every channel of its
// slots: a (slot, head) logit is finished in-lane, no cross-wave sums.  W fragments are those of the weight padded to [H Cp, 16 KS_T].
//
// The ring's lead is what the kernel's speed hangs on: an LDS-DMA request takes ~2.7 us (~5 800 cycles) to land with a CU's ring
// in flight, a tile's products ~3 650, and vector-memory operations retire IN ORDER per wave: a wave that both
// requests fragments and gathers its slots' x_l / x_r rows drains its requests every time it waits for a gather -- once per tile,
// whatever the order (the first form of this kernel: 800 us, 2 000 cycles of every tile spent waiting at its barrier).  So the
// requests have a wave of their OWN: wave 7 of the workgroup computes nothing, requests tile nt + 2 behind the barrier that frees
// its slot and arrives at a tile's barrier once that tile has landed (a counted wait: only the younger tile may still fly) -- two
// tile periods of lead; waves 0-6 (224 slots) never see a request, and their waits are the compiler's, for their gathers alone.
// One raw barrier per tile (csrc/isg_diag.hpp: -DISG_DIAG_STRICT turns wait and barrier into a full wait and __syncthreads()).
constexpr int ER_THREADS = 512, ER_WAVES = 8, ER_SLOTS = 32 * (ER_WAVES - 1), ER_RING = 3;
typedef __attribute__((address_space(3))) void er_lds_t;
typedef __attribute__((address_space(1))) void er_glb_t;
typedef float er_f4 __attribute__((ext_vector_type(4)));

ISG_DIAG_BUFFER(g_er_stamps)            // -DISG_DIAG builds only (tools/stamp_edge_logits_rows.py): [workgroups * 8 waves][16] int64

// F16: x_l / x_r are HALF rows (BASELINE configs[4]: fp16 feature rows, fp32 arithmetic; strides in halfs) and the edge projection
// is rounded to half before it is used, as the un-fused path stores it (isg_linear_f16x3_f16 -> isg_gatv2_mp_fwd_f16).
template <bool MASKED, int KS_T, bool F16 = false>
__global__ __launch_bounds__(ER_THREADS, 2) void gatv2_edge_logits_rows_kernel(ElArgs a) {
  typedef typename std::conditional<F16, _Float16, float>::type er_x_t;
  typedef er_x_t er_x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char er_smem[];
  constexpr int PIECES = 2 * KS_T;                   // one-KB pieces of a tile: [plane][k step] x (lane x 16 bytes)
  constexpr int TILE_B = PIECES * 1024;
  static_assert(PIECES < 64, "the counted wait below is vmcnt(PIECES)");
  float *s_cw = reinterpret_cast<float *>(er_smem + ER_RING * TILE_B);      // att [H*Cp] (zeros in the padding), w_inv [H*Cp]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, hh = lane >> 5;
  const int HC = a.H * a.Cp, NT = a.NT;
  ISG_DIAG_BEGIN()
  for (int c = tid; c < HC; c += ER_THREADS) {
    const int hd = c / a.Cp, ch = c - hd * a.Cp;
    s_cw[c] = ch < a.C ? a.att[hd * a.C + ch] : 0.f;
    s_cw[HC + c] = a.w_inv[c];
  }

  if (wave == ER_WAVES - 1) {           // ---- the requesting wave ---------------------------------------------------------
    const unsigned plane_b = (unsigned)NT * (unsigned)KS_T * 1024u;
    const unsigned char *Wb = reinterpret_cast<const unsigned char *>(a.Wf) + lane * 16;
#define ER_STAGE_W(nt_, slot_)                                                                                   \
    _Pragma("unroll") for (int c = 0; c < PIECES; ++c)                                                           \
      __builtin_amdgcn_global_load_lds((er_glb_t *)(Wb + (size_t)(c >= KS_T ? plane_b : 0u) +                    \
                                                    ((size_t)(nt_) * KS_T + (c >= KS_T ? c - KS_T : c)) * 1024u), \
                                       (er_lds_t *)(er_smem + (slot_) * TILE_B + c * 1024), 16, 0, 0);
    // (past the last tile the requests fetch it again into a slot nobody reads: the wait below is one count on every path)
    ER_STAGE_W(0, 0)
    ER_STAGE_W(min(1, NT - 1), 1)
    ISG_WAIT(0xC07F);                   // lgkmcnt(0): my att / w_inv writes
    ISG_DIAG_ADD(0)
    int slot = 2;
#pragma unroll 1
    for (int nt = 0; nt < NT; ++nt) {
      ISG_WAIT(((PIECES >> 4) << 14) | 0x0F70 | (PIECES & 15));      // vmcnt(PIECES): tile nt has landed, tile nt + 1 may still fly
      ISG_DIAG_ADD(1)
      __builtin_amdgcn_sched_barrier(0);
      ISG_BARRIER();                    // tile nt is handed over; every computing wave is done with tile nt - 1: its slot is free
      __builtin_amdgcn_sched_barrier(0);
      ISG_DIAG_ADD(2)
      ER_STAGE_W(min(nt + 2, NT - 1), slot)
      slot = slot == ER_RING - 1 ? 0 : slot + 1;
      ISG_DIAG_ADD(3)
    }
    ISG_WAIT(0x0F70);                   // nothing in flight at the end
    ISG_DIAG_DUMP(g_er_stamps, blockIdx.x * ER_WAVES + wave, 12, )
#undef ER_STAGE_W
    return;
  }

  // ---- this lane's slot: endpoints, mask, and its half of the edge row (k = 16 ks + 8 hh + 0..7) -> scale -> (hi, mid) ----------
  const int slot = blockIdx.x * ER_SLOTS + wave * 32 + fr;
  const int sl = min(slot, a.E - 1);
  const int s_node = a.src[sl], d_node = a.dst[sl], e_id = a.eid[sl];
  const int64_t xl_off = (int64_t)s_node * a.ldl, xr_off = (int64_t)d_node * a.ldr;
  float me = 1.f;
  if (MASKED) me = a.edge_mask ? a.edge_mask[e_id] : a.node_mask[s_node] * a.node_mask[d_node];
  hf16x8 ph[KS_T], pm[KS_T];
  float sinv;
  {
    const float *row = a.edge_attr + (int64_t)e_id * a.lda + 8 * hh;
    float4 r0[KS_T], r1[KS_T];
    float mx = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS_T; ++ks) {
      const int k0 = ks * 16 + 8 * hh;
      r0[ks] = k0 < a.K ? *reinterpret_cast<const float4 *>(row + ks * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      r1[ks] = k0 + 4 < a.K ? *reinterpret_cast<const float4 *>(row + ks * 16 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int ks = 0; ks < KS_T; ++ks)
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(r0[ks].x), fabsf(r0[ks].y)), fmaxf(fabsf(r0[ks].z), fabsf(r0[ks].w))),
                           fmaxf(fmaxf(fabsf(r1[ks].x), fabsf(r1[ks].y)), fmaxf(fabsf(r1[ks].z), fabsf(r1[ks].w)))));
    mx = fmaxf(mx, __shfl_xor(mx, 32));              // the row's other half sits in lane ^ 32
    float s;
    h3_scale(mx, s, sinv);
#pragma unroll
    for (int ks = 0; ks < KS_T; ++ks) {
      const float v[8] = {r0[ks].x * s, r0[ks].y * s, r0[ks].z * s, r0[ks].w * s, r1[ks].x * s, r1[ks].y * s, r1[ks].z * s, r1[ks].w * s};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ph[ks][j] = (_Float16)v[j];
        pm[ks][j] = (_Float16)(v[j] - (float)ph[ks][j]);
      }
    }
  }
  const int tph = a.Cp >> 5;
  const float slope = a.slope;
  // K = 128 (KS_T = 8) is also what the graph-tile kernels and the panel kernel above run, and a batch split between those
  // and this one must not show it (tests/test_gpu_ops.py: the tile convolution against the pair, bit for bit): the logit is then
  // summed in THEIR order -- tile nt's channels go to partial sums nt & 3 (the panel kernel's four tile-waves), added pairwise at
  // the head's end; the tile loop is unrolled by four so that nt & 3 is a constant.  Wider K has nobody to agree with: one set.
  constexpr int NP = KS_T == 8 ? 4 : 1;
  float part[NP][4];
#pragma unroll
  for (int u = 0; u < NP; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) part[u][g] = 0.f;
  ISG_WAIT(0xC07F);                     // lgkmcnt(0): my att / w_inv writes are in LDS before the first tile's barrier
  ISG_DIAG_ADD(0)

  // The two waves of a SIMD (w and w + 4) run their tile periods in OPPOSITE phase: waves 0-3 do products, then epilogue; waves
  // 4-6 first the epilogue of the PREVIOUS tile, then the products -- behind one barrier per tile every wave would otherwise be in
  // its products at the same time (the matrix core shared, the vector ALU idle) and in its epilogue at the same time (the matrix
  // core idle: 1 460 of a tile's 7 500 cycles in the first form).  No register is added: a late wave's accumulators and gathers
  // just live across the barrier instead of across nothing.
  const bool late = wave >= 4;
  er_x4 xl[4], xr[4];
  hf32x16 acc;
#define ER_EPILOGUE(U)                                                                                             \
  {                                                                                                                \
    /* the gathers are not to be touched before this point (whole 16-byte registers: no early copies of their parts either) */ \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(xl[g]), "+v"(xr[g]));                     \
    ISG_DIAG_ADD(4)                                                                                                \
    const int cb = e_cb;                                                                                           \
    /* ALL of the tile's att / w_inv in one go (one LDS round trip under the other waves' fragment reads, not four); the      */ \
    /* arithmetic is SCALAR fp32 (the file is built with -fno-slp-vectorize): beside another wave's MFMAs a packed fp32      */ \
    /* operation costs more issue time than the two scalar ones it replaces (MI355X_MICROARCH.md, constants table)           */ \
    er_f4 at4[4], wi4[4];                                                                                          \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
      at4[g] = *reinterpret_cast<const er_f4 *>(&s_cw[cb + 8 * g]);                                                \
      wi4[g] = *reinterpret_cast<const er_f4 *>(&s_cw[HC + cb + 8 * g]);                                           \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
        float e = (acc[g * 4 + j] * sinv) * wi4[g][j];        /* both scales are powers of two: exact */           \
        if (F16) e = (float)(_Float16)e;                      /* e_proj as the un-fused path stores it */          \
        float z = ((float)xr[g][j] + (float)xl[g][j]) + e;                                                         \
        if (MASKED) z *= me;                                                                                       \
        z = leaky(z, slope);                                                                                       \
        if (MASKED) z *= me;                                                                                       \
        part[U][g] = fmaf(z, at4[g][j], part[U][g]);                                                               \
      }                                                                                                            \
    }                                                                                                              \
    e_cb += 32;                                                                                                    \
    if (++e_tin == tph) {                    /* the head's last tile: its logit is complete in this lane pair */   \
      float tot[NP];                                                                                               \
      _Pragma("unroll") for (int u = 0; u < NP; ++u) {                                                             \
        const float mine = (part[u][0] + part[u][1]) + (part[u][2] + part[u][3]);                                  \
        tot[u] = mine + __shfl_xor(mine, 32);                                                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) part[u][g] = 0.f;                                            \
      }                                                                                                            \
      const float sum = NP == 4 ? (tot[0] + tot[1 % NP]) + (tot[2 % NP] + tot[3 % NP]) : tot[0];                   \
      if (hh == 0 && slot < a.E) a.logits[(int64_t)slot * a.H + e_hd] = sum;                                       \
      e_tin = 0;                                                                                                   \
      ++e_hd;                                                                                                      \
    }                                                                                                              \
    ISG_DIAG_KEEP2(part[0][0], part[0][3])                                                                         \
    ISG_DIAG_ADD(5)                                                                                                \
  }
  int rslot = 0;
  // running state instead of a division per tile: the tile's place in its head (products / epilogue), the lane's x_l / x_r
  // pointers at the tile's first channel, the epilogue's place in att / w_inv
  const er_x_t *pl = reinterpret_cast<const er_x_t *>(a.x_l) + xl_off + 4 * hh, *pr = reinterpret_cast<const er_x_t *>(a.x_r) + xr_off + 4 * hh;
  int p_tin = 0, e_tin = 0, e_hd = 0, e_cb = 4 * hh;
  // one tile's gathers and products
#define ER_PRODUCTS()                                                                                              \
  {                                                                                                                \
    /* UNCONDITIONAL gathers (a branch per group makes the compiler wait for each group inside its branch).  A tile inside   */ \
    /* the head: one address per operand, the groups at constant offsets; the head's last tile (wave-uniform): the padding's */ \
    /* channels read the head's last four, and att is zero there                                                            */ \
    if (p_tin * 32 + 32 <= a.C) {                                                                                  \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                              \
        xl[g] = *reinterpret_cast<const er_x4 *>(pl + 8 * g);                                                      \
        xr[g] = *reinterpret_cast<const er_x4 *>(pr + 8 * g);                                                      \
      }                                                                                                            \
    } else {                                                                                                       \
      const int room = a.C - 4 - (p_tin * 32 + 4 * hh);      /* floats from this lane's first channel to the head's last four */ \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                              \
        xl[g] = *reinterpret_cast<const er_x4 *>(pl + min(8 * g, room));                                           \
        xr[g] = *reinterpret_cast<const er_x4 *>(pr + min(8 * g, room));                                           \
      }                                                                                                            \
    }                                                                                                              \
    if (++p_tin == tph) {                                                                                          \
      p_tin = 0;                                                                                                   \
      pl += a.hsl - (tph - 1) * 32;                                                                                \
      pr += a.hsr - (tph - 1) * 32;                                                                                \
    } else {                                                                                                       \
      pl += 32;                                                                                                    \
      pr += 32;                                                                                                    \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);        /* (the scheduler would sink the gathers below the products, or add them up in front) */ \
    ISG_DIAG_ADD(2)                                                                                                \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                                   \
    const unsigned char *wt = er_smem + rslot * TILE_B + lane * 16;                                                \
    rslot = rslot == ER_RING - 1 ? 0 : rslot + 1;                                                                  \
    hf16x8 wq[2][2];          /* [stage][plane] */                                                                 \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) wq[0][q] = *reinterpret_cast<const hf16x8 *>(wt + q * (KS_T * 1024)); \
    _Pragma("unroll") for (int ks = 0; ks < KS_T; ++ks) {                                                          \
      if (ks + 1 < KS_T) {                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                              \
          wq[(ks + 1) & 1][q] = *reinterpret_cast<const hf16x8 *>(wt + q * (KS_T * 1024) + (ks + 1) * 1024);       \
      }                                                                                                            \
      /* transposed product: W fragment = A operand (rows = channels), the edge rows = B operand (columns = slots) */ \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks & 1][0], pm[ks], acc, 0, 0, 0);                           \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks & 1][1], ph[ks], acc, 0, 0, 0);                           \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[ks & 1][0], ph[ks], acc, 0, 0, 0);                           \
    }                                                                                                              \
    ISG_DIAG_KEEP2(acc[0], acc[15])                                                                                \
    ISG_DIAG_ADD(3)                                                                                                \
  }
#define ER_TILE_BARRIER()                     /* tile nt's fragments are in ring slot nt % 3 */                    \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  ISG_BARRIER();                                                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  ISG_DIAG_ADD(1)
  if constexpr (NP == 1) {
    if (!late) {              // (two loops, not one with two branches: each gets its own register allocation)
#pragma unroll 1
      for (int nt = 0; nt < NT; ++nt) {
        ER_TILE_BARRIER()
        ER_PRODUCTS()
        ER_EPILOGUE(0)
      }
    } else {
      ER_TILE_BARRIER()
      ER_PRODUCTS()
#pragma unroll 1
      for (int nt = 1; nt < NT; ++nt) {
        ER_TILE_BARRIER()
        ER_EPILOGUE(0)
        ER_PRODUCTS()
      }
      ER_EPILOGUE(0)
    }
  } else {
    if (!late) {
#pragma unroll 1
      for (int nt = 0; nt < NT; nt += 4) {
        { ER_TILE_BARRIER() ER_PRODUCTS() ER_EPILOGUE(0) }
        if (nt + 1 < NT) { ER_TILE_BARRIER() ER_PRODUCTS() ER_EPILOGUE(1) }
        if (nt + 2 < NT) { ER_TILE_BARRIER() ER_PRODUCTS() ER_EPILOGUE(2) }
        if (nt + 3 < NT) { ER_TILE_BARRIER() ER_PRODUCTS() ER_EPILOGUE(3) }
      }
    } else {                  // a late wave's epilogue is the PREVIOUS tile's: (nt - 1) & 3
#pragma unroll 1
      for (int nt = 0; nt < NT; nt += 4) {
        { ER_TILE_BARRIER() if (nt > 0) ER_EPILOGUE(3) ER_PRODUCTS() }
        if (nt + 1 < NT) { ER_TILE_BARRIER() ER_EPILOGUE(0) ER_PRODUCTS() }
        if (nt + 2 < NT) { ER_TILE_BARRIER() ER_EPILOGUE(1) ER_PRODUCTS() }
        if (nt + 3 < NT) { ER_TILE_BARRIER() ER_EPILOGUE(2) ER_PRODUCTS() }
      }
      switch ((NT - 1) & 3) {
        case 0: ER_EPILOGUE(0) break;
        case 1: ER_EPILOGUE(1) break;
        case 2: ER_EPILOGUE(2) break;
        default: ER_EPILOGUE(3) break;
      }
    }
  }
#undef ER_PRODUCTS
#undef ER_TILE_BARRIER
#undef ER_EPILOGUE
  ISG_DIAG_DUMP(g_er_stamps, blockIdx.x * ER_WAVES + wave, 12, )
}

}  // namespace isg

using namespace isg;

ISG_DIAG_SETTER(isg_er_set_stamp_buffer, g_er_stamps)

static int edge_logits(const float *edge_attr, int32_t lda, const uint16_t *w_frag, const float *w_inv_scale,
                       const void *x_l, int32_t ldl, int64_t head_stride_l, const void *x_r, int32_t ldr,
                       int64_t head_stride_r, const float *att,
                       const int32_t *eid, const int32_t *src, const int32_t *dst, const float *edge_mask,
                       const float *node_mask, float *logits, int64_t E, int32_t H, int32_t C, int32_t K,
                       float negative_slope, void *stream, bool f16) {
  if (head_stride_l == 0) head_stride_l = C;
  if (head_stride_r == 0) head_stride_r = C;
  if (E < 0 || H <= 0 || C <= 0 || K <= 0 || lda < K || ldl < C || head_stride_l < C) return ISG_EINVAL;
  if (ldr < C || head_stride_r < C) return ISG_EINVAL;
  if ((head_stride_l == C && ldl < H * C) || (head_stride_r == C && ldr < H * C)) return ISG_EINVAL;
  if (E == 0) return ISG_OK;
  if (!edge_attr || !w_frag || !w_inv_scale || !x_l || !x_r || !att || !eid || !src || !dst || !logits) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  const int Cp = (C + 31) / 32 * 32;             // heads padded to whole 32-channel tiles (w_frag / w_inv_scale are the PADDED weight's)
  if ((C & 3) != 0 || H > 32 || K > 304 || (K & 3) != 0 || (lda & 3) != 0 || (ldl & 3) != 0 || (ldr & 3) != 0 || (head_stride_l & 3) != 0 || (head_stride_r & 3) != 0 || mis(edge_attr) ||
      mis(x_l) || mis(x_r) || mis(att) || mis(w_inv_scale) || H * Cp > 2048 || E >= (1ll << 31) - EL_BM)
    return ISG_EUNSUPPORTED;
  // every field named, in declaration order: -Werror=missing-field-initializers (HIP_FLAGS) refuses a field left out
  ElArgs a = {
      .edge_attr = edge_attr, .Wf = reinterpret_cast<const _Float16 *>(w_frag), .w_inv = w_inv_scale,
      .x_l = static_cast<const float *>(x_l), .x_r = static_cast<const float *>(x_r),   // (half rows when f16: the kernel casts)
      .att = att, .eid = eid, .src = src, .dst = dst, .edge_mask = edge_mask, .node_mask = node_mask, .logits = logits,
      .E = (int)E, .H = H, .C = C, .K = K, .KS = (K + 15) / 16, .NT = H * Cp / 32, .lda = lda, .ldl = ldl, .ldr = ldr,
      .Cp = Cp, .LD = EL_KC + 8, .hsl = head_stride_l, .hsr = head_stride_r, .slope = negative_slope};
  if (!a.edge_attr || !a.Wf || !a.w_inv || !a.x_l || !a.x_r || !a.att || !a.eid || !a.src || !a.dst || !a.logits)
    return ISG_EINVAL;                         // the struct the kernel dereferences, not the parameters it was filled from
  hipStream_t st = as_stream(stream);
  const bool masked = edge_mask || node_mask;
  // the rows kernel: w_frag is the split of the weight padded to [H * Cp, 16 KST]: 304 columns (19 k steps) for K > 128, and for
  // K = 128 exactly its own 128 (8 k steps; a narrower K stays on the panel kernel, whose k loop is as long as K)
  auto rows = [&](auto kst) -> int {
    constexpr int KST = decltype(kst)::value;
    a.KS = KST;
    const unsigned grid = (unsigned)((E + ER_SLOTS - 1) / ER_SLOTS);
    const size_t dyn = (size_t)ER_RING * (2 * KST * 1024) + (size_t)2 * H * Cp * sizeof(float);
    // (the attribute is set once per kernel: to the most any shape asks for -- H Cp <= 2048 -- not to the first caller's size)
    constexpr int dyn_max = ER_RING * (2 * KST * 1024) + 2 * 2048 * (int)sizeof(float);
#define ER_LAUNCH(M_, F_)                                                                                          \
    {                                                                                                              \
      if (!dyn_lds_ok<&gatv2_edge_logits_rows_kernel<M_, KST, F_>>(dyn_max)) return ISG_EUNSUPPORTED;             \
      gatv2_edge_logits_rows_kernel<M_, KST, F_><<<grid, ER_THREADS, dyn, st>>>(a);                               \
    }
    if (masked && f16) ER_LAUNCH(true, true)
    else if (masked) ER_LAUNCH(true, false)
    else if (f16) ER_LAUNCH(false, true)
    else ER_LAUNCH(false, false)
#undef ER_LAUNCH
    return check_launch();
  };
  if (K > EL_KC) return rows(std::integral_constant<int, 19>{});
  if (K == EL_KC) return rows(std::integral_constant<int, 8>{});        // 177 against 228 us on the panel kernel (BASELINE configs[1] topology)
  if (f16) return ISG_EUNSUPPORTED;            // half rows: the rows kernel only (K >= 128)
  const unsigned grid = (unsigned)((E + EL_BM - 1) / EL_BM);
  const size_t dyn = (size_t)2 * EL_BM * a.LD * 2 + ((size_t)2 * H * Cp + (size_t)4 * EL_BM * H) * sizeof(float);
  // the attribute is set to the most any accepted shape asks for (H Cp <= 2048, H <= 32: 83,968 bytes), not to the first caller's
  // size -- a first launch at H = 4 (43 KB) must not leave H = 16 (67.6 KB, beyond the 64 KB default limit) unlaunchable
  constexpr int dyn_max = 2 * EL_BM * (EL_KC + 8) * 2 + (2 * 2048 + 4 * EL_BM * 32) * (int)sizeof(float);
  if (dyn > (size_t)dyn_max) return ISG_EUNSUPPORTED;
  if (masked) {
    if (!dyn_lds_ok<&gatv2_edge_logits_kernel<true, 1>>(dyn_max)) return ISG_EUNSUPPORTED;
    gatv2_edge_logits_kernel<true, 1><<<grid, EL_THREADS, dyn, st>>>(a);
  } else {
    if (!dyn_lds_ok<&gatv2_edge_logits_kernel<false, 1>>(dyn_max)) return ISG_EUNSUPPORTED;
    gatv2_edge_logits_kernel<false, 1><<<grid, EL_THREADS, dyn, st>>>(a);
  }
  return check_launch();
}

extern "C" int isg_gatv2_edge_logits(const float *edge_attr, int32_t lda, const uint16_t *w_frag, const float *w_inv_scale,
                                     const float *x_l, int32_t ldl, int64_t head_stride_l, const float *x_r, int32_t ldr,
                                     int64_t head_stride_r, const float *att,
                                     const int32_t *eid, const int32_t *src, const int32_t *dst, const float *edge_mask,
                                     const float *node_mask, float *logits, int64_t E, int32_t H, int32_t C, int32_t K,
                                     float negative_slope, void *stream) {
  return edge_logits(edge_attr, lda, w_frag, w_inv_scale, x_l, ldl, head_stride_l, x_r, ldr, head_stride_r, att, eid, src, dst,
                     edge_mask, node_mask, logits, E, H, C, K, negative_slope, stream, false);
}

// The same with x_l / x_r as HALF rows (strides and head strides in halfs), BASELINE configs[4]'s feature storage: the edge
// projection is rounded to half before it enters the logit, as isg_linear_f16x3_f16 stores it for isg_gatv2_mp_fwd_f16.  K >= 128.
extern "C" int isg_gatv2_edge_logits_f16(const float *edge_attr, int32_t lda, const uint16_t *w_frag, const float *w_inv_scale,
                                         const uint16_t *x_l, int32_t ldl, int64_t head_stride_l, const uint16_t *x_r,
                                         int32_t ldr, int64_t head_stride_r, const float *att, const int32_t *eid,
                                         const int32_t *src, const int32_t *dst, const float *edge_mask, const float *node_mask,
                                         float *logits, int64_t E, int32_t H, int32_t C, int32_t K, float negative_slope,
                                         void *stream) {
  return edge_logits(edge_attr, lda, w_frag, w_inv_scale, x_l, ldl, head_stride_l, x_r, ldr, head_stride_r, att, eid, src, dst,
                     edge_mask, node_mask, logits, E, H, C, K, negative_slope, stream, true);
}
