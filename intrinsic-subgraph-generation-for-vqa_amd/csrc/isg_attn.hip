// Multi-head attention over SHORT sequences: the question encoder / program decoder of ISubGVQA
// (ISubGVQA/models/question_encoder.py:20-38, question_decoder.py:25-71: nn.TransformerEncoder / Decoder, d = 512,
// 8 heads of 64, T <= 77 CLIP tokens, 4 instruction queries).  softmax(Q K^T / sqrt(hd) + key_bias) V per (batch item, head).
//
// The reference hands torch's MultiheadAttention a FLOAT key-padding mask, which torch ADDS to the scores (+1 on real
// tokens, pads attended: SURVEY App. B Q5); `key_bias` is that additive term.  A question's K and V for one head are
// T x 64 floats: they live in LDS for the whole workgroup, there is no tiling, no online softmax and nothing to
// re-read -- one workgroup per (batch item, head), one wave per query row:
//   scores   lane = key (two keys per lane beyond 64): dot over the head dimension from LDS (K rows padded by one float:
//            conflict-free), scaled, biased
//   softmax  wave max / libm expf / wave sum / IEEE divide
//   P V      lane = channel: sum over the keys of p_s V[s][c] in key order
// Rows follow torch's [T, B, D] layout (row = t * B + b), so q / k / v may be column slices of the fused in_proj output.
#include "isg_f16x3.hpp"

namespace isg {

struct MhaArgs {
  const float *q, *k, *v, *key_bias;
  float *out, *rowmax;     // rowmax [Tq*B, H] or NULL: max |out| per (row, head), the row scales of the Linear that reads out
  int B, H, hd, Tq, Tk, ldq, ldk, ldv, ldo;
  float scale;
  _Float16 *planes;        // ROWS form: the result as the planes32 operand of isg_linear_h3p (out_proj), or NULL
  float *planes_inv;
};

// PARTS lanes share one key's dot product (the head dimension in PARTS interleaved slices, combined by a DPP butterfly): with
// lane = key, a 12-token question kept 12 of 64 lanes busy for 64 dependent FMAs per query row -- the kernel was bound by that
// chain and by one global load of the query row per round (152 us for 49 152 x 8 rows: 2.6 TB/s of a 400 MB pass).  Q, K and
// V of the head now arrive together (16-byte loads), a query row costs 64 / PARTS FMAs per lane.
// ROWS: one workgroup per batch item walks ALL heads and assembles the item's result rows [Tq][H * hd] in LDS, so that a row's
// largest magnitude over all heads is known where the row is written: the result leaves as planes32 (and as fp32 rows where `out`
// is given) and out_proj needs no isg_split_planes32 pass over it -- ten such passes per full-model step (36 / 21 us each at 4096
// questions).  Needs Tq * H * hd more floats of LDS: 12-token questions and the decoder's 4 queries, not CLIP's 77.
// NW waves per workgroup, one query row per wave and round: a row is a chain of LDS round trips (scores -> strip -> softmax ->
// strip -> P V), so a 12-token head on 4 waves was three such chains in a row per wave; the ROWS form runs 12 waves for it.
template <int PARTS, bool ROWS, int NW = 4>
__global__ __launch_bounds__(64 * NW) void mha_small_kernel(MhaArgs a) {
  constexpr int NT = 64 * NW;
  extern __shared__ float smem[];
  const int hd = a.hd, Tk = a.Tk, kp = hd + 4;       // K rows padded by one float4: conflict-free 16-byte reads down a column
  float *Ks = smem;                       // [Tk][hd + 4]
  float *Vs = Ks + (size_t)Tk * kp;       // [Tk][hd]
  float *Qs = Vs + (size_t)Tk * hd;       // [Tq][hd]
  float *ps = Qs + (size_t)a.Tq * hd;     // [NW][128]
  float *Cs = ps + NW * 128;              // ROWS: [Tq][H * hd]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = ROWS ? blockIdx.x : blockIdx.x / a.H;
  const int h4 = hd >> 2, D = a.H * hd;
  float *pw = ps + wave * 128;
  constexpr int KPL = 64 / PARTS;          // keys per wave pass
  const int part = lane % PARTS, kslot = lane / PARTS;
  // ROWS: the NEXT head's Q / K / V are requested into registers before this head's rows are computed (up to NPF float4 per
  // thread and operand: a 12-token head is 192 float4 per operand over 256 threads) and stored to LDS behind the barrier that
  // ends the head -- eight heads in a row would otherwise pay eight exposed round trips to memory per workgroup.
  constexpr int NPF = NW >= 8 ? 1 : 2;
  const bool pf = ROWS && Tk * h4 <= NPF * NT && a.Tq * h4 <= NPF * NT;
  float4 rk[NPF], rv[NPF], rq[NPF];
  // (macros, not lambdas: register arrays captured by reference end up in scratch memory)
#define MHA_REQUEST(hh)                                                                                            \
  {                                                                                                                \
    const int c0_ = (hh) * hd;                                                                                     \
    _Pragma("unroll") for (int u = 0; u < NPF; ++u) {                                                              \
      const int idx = tid + NT * u;                                                                                \
      rk[u] = rv[u] = rq[u] = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
      if (idx < Tk * h4) {                                                                                         \
        const int s_ = idx / h4, c = idx - s_ * h4;                                                                \
        const size_t row = (size_t)s_ * a.B + b;                                                                   \
        rk[u] = *reinterpret_cast<const float4 *>(a.k + row * a.ldk + c0_ + 4 * c);                                \
        rv[u] = *reinterpret_cast<const float4 *>(a.v + row * a.ldv + c0_ + 4 * c);                                \
      }                                                                                                            \
      if (idx < a.Tq * h4) {                                                                                       \
        const int t = idx / h4, c = idx - t * h4;                                                                  \
        rq[u] = *reinterpret_cast<const float4 *>(a.q + ((size_t)t * a.B + b) * a.ldq + c0_ + 4 * c);             \
      }                                                                                                            \
    }                                                                                                              \
  }
#define MHA_LAND()                                                                                                 \
  _Pragma("unroll") for (int u = 0; u < NPF; ++u) {                                                                \
    const int idx = tid + NT * u;                                                                                  \
    if (idx < Tk * h4) {                                                                                           \
      const int s_ = idx / h4, c = idx - s_ * h4;                                                                  \
      *reinterpret_cast<float4 *>(Ks + s_ * kp + 4 * c) = rk[u];                                                   \
      *reinterpret_cast<float4 *>(Vs + s_ * hd + 4 * c) = rv[u];                                                   \
    }                                                                                                              \
    if (idx < a.Tq * h4) {                                                                                         \
      const int t = idx / h4, c = idx - t * h4;                                                                    \
      *reinterpret_cast<float4 *>(Qs + t * hd + 4 * c) = rq[u];                                                    \
    }                                                                                                              \
  }
  if (pf) MHA_REQUEST(0)
  for (int h = ROWS ? 0 : blockIdx.x - b * a.H, hend = ROWS ? a.H : h + 1; h < hend; ++h) {
    const int col0 = h * hd;
    if (pf) {
      MHA_LAND()
      if (h + 1 < hend) MHA_REQUEST(h + 1)
    } else {
      for (int idx = tid; idx < Tk * h4; idx += NT) {
        const int s = idx / h4, c = idx - s * h4;
        const size_t row = (size_t)s * a.B + b;
        *reinterpret_cast<float4 *>(Ks + s * kp + 4 * c) = *reinterpret_cast<const float4 *>(a.k + row * a.ldk + col0 + 4 * c);
        *reinterpret_cast<float4 *>(Vs + s * hd + 4 * c) = *reinterpret_cast<const float4 *>(a.v + row * a.ldv + col0 + 4 * c);
      }
      for (int idx = tid; idx < a.Tq * h4; idx += NT) {
        const int t = idx / h4, c = idx - t * h4;
        *reinterpret_cast<float4 *>(Qs + t * hd + 4 * c) =
            *reinterpret_cast<const float4 *>(a.q + ((size_t)t * a.B + b) * a.ldq + col0 + 4 * c);
      }
    }
    __syncthreads();
    for (int tq = wave; tq < a.Tq; tq += NW) {
      const size_t qrow = (size_t)tq * a.B + b;
      const float *qw = Qs + tq * hd;
      float mx = -INFINITY;
      for (int s0 = 0; s0 < Tk; s0 += KPL) {       // scores of KPL keys at a time
        const int s = s0 + kslot;
        float dot = 0.f;
        if (s < Tk) {
          const float *kr = Ks + s * kp;
          for (int c = part * 4; c < hd; c += 4 * PARTS) {          // this lane's interleaved float4 slices of the head
            const float4 qv = *reinterpret_cast<const float4 *>(qw + c), kv = *reinterpret_cast<const float4 *>(kr + c);
            dot = fmaf(qv.x, kv.x, dot); dot = fmaf(qv.y, kv.y, dot); dot = fmaf(qv.z, kv.z, dot); dot = fmaf(qv.w, kv.w, dot);
          }
        }
        if (PARTS >= 2) dot += dpp_mov<ISG_DPP_XOR1>(dot);
        if (PARTS >= 4) dot += dpp_mov<ISG_DPP_XOR2>(dot);
        if (s < Tk && part == 0) {
          dot *= a.scale;
          if (a.key_bias) dot += a.key_bias[(size_t)b * Tk + s];
          pw[s] = dot;
          mx = fmaxf(mx, dot);
        }
      }
      mx = wave_max(mx);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float e0 = 0.f, e1 = 0.f;
      if (lane < Tk) e0 = expf(pw[lane] - mx);
      if (64 + lane < Tk) e1 = expf(pw[64 + lane] - mx);
      const float den = wave_sum(e0 + e1);
      __builtin_amdgcn_wave_barrier();
      if (lane < Tk) pw[lane] = e0 / den;
      if (64 + lane < Tk) pw[64 + lane] = e1 / den;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float o = 0.f;
      if (lane < hd) {
        for (int s = 0; s < Tk; ++s) o = fmaf(pw[s], Vs[s * hd + lane], o);
        if (ROWS) Cs[tq * D + col0 + lane] = o;
        else a.out[qrow * a.ldo + col0 + lane] = o;
      }
      if (!ROWS && a.rowmax) {
        const float m = wave_max(fabsf(o));
        if (lane == 0) a.rowmax[qrow * a.H + h] = m;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (ROWS) __syncthreads();              // the next head's operands overwrite this one's; after the last head: the rows are whole
  }
  if constexpr (ROWS) {
    const int KT = (D + 31) >> 5, nc = D >> 2;
    for (int tq = wave; tq < a.Tq; tq += NW) {
      const size_t qrow = (size_t)tq * a.B + b;
      const float4 *c4 = reinterpret_cast<const float4 *>(Cs + (size_t)tq * D);
      float4 *o4 = a.out ? reinterpret_cast<float4 *>(a.out + qrow * a.ldo) : nullptr;
      planes32_row<64, 0>(lane, nc, KT, a.planes + qrow * KT * 64, a.planes_inv + qrow, [&](int c) { return c4[c]; },
                          [&](int c, float4 t) { if (o4) o4[c] = t; });
    }
  }
}

#undef MHA_REQUEST
#undef MHA_LAND

// out = LayerNorm(x + r) over the last dimension (r optional), torch.nn.LayerNorm's arithmetic order
// ((v - mean) * rstd * gamma + beta, biased variance, fp32): the post-norm steps of nn.TransformerEncoderLayer /
// DecoderLayer (question_encoder.py:20-38, question_decoder.py:25-71) with the residual add folded in, and the row's
// largest |out| written beside it -- the row scale of the fp16 three-product Linear that reads `out` next, so that
// Linear needs no pass of its own over its input.  Wave per row, the row in registers (NV float4 per lane), mean and
// centred variance as two wave sums.
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ x, const float *__restrict__ r,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps,
                                                            float *__restrict__ out, float *__restrict__ rowmax, int M,
                                                            int D, int ldx, int ldr, int ldo,
                                                            _Float16 *__restrict__ planes, float *__restrict__ pinv) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = D >> 2;
  float4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nv) {
      v[i] = reinterpret_cast<const float4 *>(x + (int64_t)row * ldx)[c];
      if (r) {
        const float4 t = reinterpret_cast<const float4 *>(r + (int64_t)row * ldr)[c];
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (lane + 64 * i < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4 g = reinterpret_cast<const float4 *>(gamma)[c];
      const float4 bb = beta ? reinterpret_cast<const float4 *>(beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + bb.x;
      o.y = (v[i].y - mean) * rstd * g.y + bb.y;
      o.z = (v[i].z - mean) * rstd * g.z + bb.z;
      o.w = (v[i].w - mean) * rstd * g.w + bb.w;
      reinterpret_cast<float4 *>(out + (int64_t)row * ldo)[c] = o;
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
      v[i] = o;
    }
  }
  if (rowmax || planes) mx = wave_max(mx);
  if (rowmax && lane == 0) rowmax[row] = mx;
  if (planes) {
    // the row as the planes32 operand of isg_linear_h3p (csrc/isg_gemm_h3p.hip): the wave holds the whole row and knows its
    // exact maximum, so the Linears that read the result (in_proj, the cross-attention projections, linear1) need no split
    // pass: k-tile kt of a row = [hi 32 | mid 32] fp16, D % 32 == 0
    typedef __attribute__((ext_vector_type(4))) _Float16 ln_h4;
    const int e = (int)((__float_as_uint(mx) >> 23) & 255u);
    float sc = 1.f, inv = 1.f;
    if (e >= 14 && e != 255) { sc = __uint_as_float((unsigned)(127 + 13 + 127 - e) << 23); inv = __uint_as_float((unsigned)(e - 13) << 23); }
    if (lane == 0) pinv[row] = inv;
    _Float16 *p = planes + (int64_t)row * (D * 2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) {
        const float4 o = v[i];
        const float a0 = o.x * sc, a1 = o.y * sc, a2 = o.z * sc, a3 = o.w * sc;
        const ln_h4 hi = {(_Float16)a0, (_Float16)a1, (_Float16)a2, (_Float16)a3};
        const ln_h4 mid = {(_Float16)(a0 - (float)hi[0]), (_Float16)(a1 - (float)hi[1]), (_Float16)(a2 - (float)hi[2]),
                           (_Float16)(a3 - (float)hi[3])};
        _Float16 *d = p + (c >> 3) * 64 + (c & 7) * 4;
        *reinterpret_cast<ln_h4 *>(d) = hi;
        *reinterpret_cast<ln_h4 *>(d + 32) = mid;
      }
    }
  }
}

}  // namespace isg

using namespace isg;

extern "C" int isg_add_layernorm(const float *x, int32_t ldx, const float *r, int32_t ldr, const float *gamma,
                                 const float *beta, float eps, float *out, int32_t ldo, float *rowmax, int64_t M,
                                 int32_t D, uint16_t *planes, float *planes_inv, void *stream) {
  if (M < 0 || D <= 0 || ldx < D || ldo < D || (r && ldr < D)) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!x || !gamma || !out) return ISG_EINVAL;
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if ((D & 3) || D > 2048 || (ldx & 3) || (ldo & 3) || (r && (ldr & 3)) || mis(x) || mis(out) || (r && mis(r)) ||
      mis(gamma) || (beta && mis(beta)) || (M + 3) / 4 >= (1ll << 31))
    return ISG_EUNSUPPORTED;
  if ((planes != nullptr) != (planes_inv != nullptr)) return ISG_EINVAL;
  if (planes && ((D & 31) || mis(planes))) return ISG_EUNSUPPORTED;
  _Float16 *pl = reinterpret_cast<_Float16 *>(planes);
  const unsigned grid = (unsigned)((M + 3) / 4);
  hipStream_t st = as_stream(stream);
#define ISG_LN(NV_) add_layernorm_kernel<NV_><<<grid, 256, 0, st>>>(x, r, gamma, beta, eps, out, rowmax, (int)M, D, ldx, ldr, ldo, pl, planes_inv)
  if (D <= 256) ISG_LN(1);
  else if (D <= 512) ISG_LN(2);
  else if (D <= 1024) ISG_LN(4);
  else ISG_LN(8);
#undef ISG_LN
  return check_launch();
}

extern "C" int isg_mha_small(const float *q, int32_t ldq, const float *k, int32_t ldk, const float *v, int32_t ldv,
                             const float *key_bias, float *out, int32_t ldo, float *rowmax, int64_t B, int32_t H, int32_t hd,
                             int32_t Tq, int32_t Tk, uint16_t *planes, float *planes_inv, void *stream) {
  if (B < 0 || H <= 0 || hd <= 0 || Tq < 0 || Tk <= 0) return ISG_EINVAL;
  if (B == 0 || Tq == 0) return ISG_OK;
  if (!q || !k || !v || (!out && !planes) || (!planes != !planes_inv)) return ISG_EINVAL;
  if (hd > 64 || Tk > 128 || B * H >= (1ll << 31)) return ISG_EUNSUPPORTED;
  if (ldq < H * hd || ldk < H * hd || ldv < H * hd || (out && ldo < H * hd)) return ISG_EINVAL;
  MhaArgs a{q, k, v, key_bias, out, rowmax, (int)B, H, hd, Tq, Tk, ldq, ldk, ldv, ldo, (float)(1.0 / sqrt((double)hd)),
            reinterpret_cast<_Float16 *>(planes), planes_inv};
  auto mis = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if ((hd & 3) || (ldq & 3) || (ldk & 3) || (ldv & 3) || mis(q) || mis(k) || mis(v)) return ISG_EUNSUPPORTED;
  const int nw = !planes || Tq <= 4 ? 4 : Tq <= 8 ? 8 : 12;      // waves per workgroup: a query row each (ROWS form)
  size_t lds = ((size_t)Tk * (2 * hd + 4) + (size_t)Tq * hd + nw * 128) * sizeof(float);
  if (planes) lds += (size_t)Tq * H * hd * sizeof(float);
  if (lds > 64 * 1024) return ISG_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (planes) {                  // all heads of a batch item in one workgroup, the rows as planes32 (+ fp32 rows if `out`)
    if (mis(planes) || (out && ((ldo & 3) || mis(out))) || rowmax) return ISG_EUNSUPPORTED;
    const unsigned grid = (unsigned)B;
#define ISG_MHA_ROWS(NW_)                                                                                          \
  do {                                                                                                             \
    if (Tk <= 16 && (hd & 15) == 0) mha_small_kernel<4, true, NW_><<<grid, 64 * NW_, lds, st>>>(a);                \
    else if (Tk <= 32 && (hd & 7) == 0) mha_small_kernel<2, true, NW_><<<grid, 64 * NW_, lds, st>>>(a);            \
    else mha_small_kernel<1, true, NW_><<<grid, 64 * NW_, lds, st>>>(a);                                           \
  } while (0)
    if (nw == 4) ISG_MHA_ROWS(4); else if (nw == 8) ISG_MHA_ROWS(8); else ISG_MHA_ROWS(12);
#undef ISG_MHA_ROWS
    return check_launch();
  }
  const unsigned grid = (unsigned)(B * H);
  if (Tk <= 16 && (hd & 15) == 0) mha_small_kernel<4, false><<<grid, 256, lds, st>>>(a);
  else if (Tk <= 32 && (hd & 7) == 0) mha_small_kernel<2, false><<<grid, 256, lds, st>>>(a);
  else mha_small_kernel<1, false><<<grid, 256, lds, st>>>(a);
  return check_launch();
}
