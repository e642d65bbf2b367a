// fp32 linear layers on the bf16 matrix cores, row-panel form: D[M,N] = act(A[M,K] . W[N,K]^T + bias)
//
// Same arithmetic as isg_gemm.hip (x = x1 + x2 + x3 exactly in bf16, the six products with i + j <= 4, fp32 accumulate,
// small terms first), different decomposition.  The tile kernel there re-reads and re-SPLITS an A tile once per 128
// output columns and has only K / 32 k-steps to hide its load -> split -> MFMA -> store phases behind each other; at the
// projections that dominate the step (K = 128, N = 512 / 1024: lin_edge, lin_l | lin_r) it ran at 0.27 of either of its
// rooflines (profiles/r01_e).  Here:
//
//   workgroup  = one PANEL of 64 rows, 4 waves (one per SIMD); 2 workgroups per CU, whose phases interleave
//   A          fp32 rows -> registers -> three bf16 planes in LDS, ONCE per panel when K <= 128 (the panel then serves
//              every output column: A-stationary), once per (column pass, 128-wide k chunk) otherwise; the next chunk's
//              global loads are in flight during the current chunk's MFMAs
//   W          never touches LDS: the planes are stored FRAGMENT-MAJOR by isg_split_bf16x3_frag,
//              Wf[plane][n / 32][k / 16][lane][8], so that the B operand of one v_mfma_f32_32x32x16_bf16 (lane l holds
//              W[n0 + (l & 31)][k0 + 8 (l >> 5) .. + 7]) is ONE contiguous 1 KB global load straight into the MFMA
//              registers; W is a few hundred KB and lives in L2
//   wave       owns WTN (1 or 2) 32-column subtiles x both 32-row halves of the panel per pass: 6 A fragments from LDS
//              (shared by the 4 waves) + 3 WTN B fragments from L2 feed 12 WTN MFMAs per 16-deep k-step; both fragment
//              sets are double buffered one k-step ahead
//   epilogue   + bias, optional exact GELU, stores straight from the accumulator layout: for a fixed register the lanes
//              0-31 / 32-63 write two whole 128-byte lines
// The accumulation order of an output element depends on nothing but K (no k rotation by row block): a row's result is
// the same wherever the row sits in a batch.
#include "isg_common.hpp"

#include <stdlib.h>

namespace isg {

typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 pbf16x4;
typedef __attribute__((ext_vector_type(16))) float pf32x16;

constexpr int PN_BM = 64;           // rows per panel
constexpr int PN_KC = 128;          // k per LDS chunk (8 k-steps of 16)
constexpr int PN_LD = PN_KC + 8;    // bf16 per LDS row: 272 bytes -> consecutive rows shift by one 16-byte slot
constexpr int PN_THREADS = 256;

__device__ __forceinline__ float pn_bf16_to_f32(__bf16 v) {
  return __uint_as_float(((unsigned)__builtin_bit_cast(unsigned short, v)) << 16);
}
__device__ __forceinline__ void pn_split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
  p1 = (__bf16)x;
  const float r1 = x - pn_bf16_to_f32(p1);
  p2 = (__bf16)r1;
  const float r2 = r1 - pn_bf16_to_f32(p2);
  p3 = (__bf16)r2;
}

// Wf[q][nt][ks][lane][j] = plane q of w[nt*32 + (lane & 31)][ks*16 + 8*(lane >> 5) + j]  (zero outside [N, K])
__global__ void split_bf16x3_frag_kernel(const float *__restrict__ w, int N, int K, int NT, int KS,
                                         __bf16 *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)NT * KS * 512;
  if (idx >= total) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const int64_t rest = idx >> 9;
  const int ks = (int)(rest % KS), nt = (int)(rest / KS);
  const int n = nt * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
  __bf16 p1 = (__bf16)0.f, p2 = p1, p3 = p1;
  if (n < N && k < K) pn_split3(w[(int64_t)n * K + k], p1, p2, p3);
  out[idx] = p1;
  out[total + idx] = p2;
  out[2 * total + idx] = p3;
}

typedef _Float16 pn_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned pn_u32x2 __attribute__((ext_vector_type(2)));
template <bool A16> struct PnRawA { typedef float4 type; };
template <> struct PnRawA<true> { typedef pn_u32x2 type; };
__device__ __forceinline__ float4 pn_cvt(const float4 &v) { return v; }
__device__ __forceinline__ float4 pn_cvt(const pn_u32x2 &v) {
  const pn_f16x4 h = __builtin_bit_cast(pn_f16x4, v);
  return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
}

typedef unsigned pn_u32x4 __attribute__((ext_vector_type(4)));

// SINGLE: K <= 128, the whole panel is staged once (A-stationary); the staging registers are then dead in the main loop
// DBG: compile-time ablation switches of the profiling build (tools/ablate_panel.py; outputs are wrong unless 0):
// 1 no B loads inside the k loop, 2 no epilogue stores, 4 no MFMA, 8 no A staging (global loads + split), 16 no A-fragment
// LDS reads inside the k loop
typedef int pn_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pn_keep(const pbf16x8 &v) { asm volatile("" ::"v"(__builtin_bit_cast(pn_i32x4, v))); }

template <int ACT, int WTN, bool A16, bool D16, bool SINGLE, int DBG = 0>
__global__ __launch_bounds__(PN_THREADS, (WTN == 1 && SINGLE) ? 3 : 2) void linear_panel_kernel(const void *__restrict__ Av,
                                                                     const __bf16 *__restrict__ Wf,
                                                                     const float *__restrict__ bias, void *__restrict__ Dv,
                                                                     int M, int N, int K, int KS, int NT, int lda, int ldd,
                                                                     int nt_store, int out_cols, int64_t out_stride) {
  typedef typename PnRawA<A16>::type RawA;
  __shared__ __attribute__((aligned(16))) __bf16 sA[3][PN_BM][PN_LD];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * PN_BM;
  const int nchunks = SINGLE ? 1 : (KS + 7) >> 3;
  const int npass = (NT + 4 * WTN - 1) / (4 * WTN);
  constexpr bool single = SINGLE;
  const int fr = lane & 31, fk = (lane >> 5) * 8;

  RawA ra[8];   // staging image of one A chunk: 64 rows x 128 k = 32 elements per thread
#define PN_LOAD_CHUNK(kc)                                                                                        \
  {                                                                                                              \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                              \
      const int i = tid + PN_THREADS * u;                                                                        \
      const int row = i >> 5, c4 = i & 31;                                                                       \
      /* never a conditional load: clamp the address, zero at store time */                                      \
      const int gr = min(m0 + row, M - 1), gk = min((kc) * PN_KC + c4 * 4, K - 4);                               \
      if constexpr (A16)                                                                                         \
        ra[u] = *reinterpret_cast<const RawA *>(reinterpret_cast<const _Float16 *>(Av) + (int64_t)gr * lda + gk); \
      else                                                                                                       \
        ra[u] = *reinterpret_cast<const RawA *>(reinterpret_cast<const float *>(Av) + (int64_t)gr * lda + gk);   \
    }                                                                                                            \
  }
#define PN_STORE_CHUNK(kc)                                                                                       \
  {                                                                                                              \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                              \
      const int i = tid + PN_THREADS * u;                                                                        \
      const int row = i >> 5, c4 = i & 31;                                                                       \
      float4 av = pn_cvt(ra[u]);                                                                                 \
      if (m0 + row >= M || (kc) * PN_KC + c4 * 4 >= K) av = make_float4(0.f, 0.f, 0.f, 0.f);                     \
      pbf16x4 p0, p1, p2;                                                                                        \
      __bf16 t0, t1, t2;                                                                                         \
      pn_split3(av.x, t0, t1, t2); p0[0] = t0; p1[0] = t1; p2[0] = t2;                                           \
      pn_split3(av.y, t0, t1, t2); p0[1] = t0; p1[1] = t1; p2[1] = t2;                                           \
      pn_split3(av.z, t0, t1, t2); p0[2] = t0; p1[2] = t1; p2[2] = t2;                                           \
      pn_split3(av.w, t0, t1, t2); p0[3] = t0; p1[3] = t1; p2[3] = t2;                                           \
      *reinterpret_cast<pbf16x4 *>(&sA[0][row][c4 * 4]) = p0;                                                    \
      *reinterpret_cast<pbf16x4 *>(&sA[1][row][c4 * 4]) = p1;                                                    \
      *reinterpret_cast<pbf16x4 *>(&sA[2][row][c4 * 4]) = p2;                                                    \
    }                                                                                                            \
  }

  if constexpr (!(DBG & 8)) PN_LOAD_CHUNK(0)
  if (single) {
    if constexpr (!(DBG & 8)) PN_STORE_CHUNK(0)
    __syncthreads();
  }

  // W fragments through one buffer descriptor: per-lane part = lane * 16 bytes (voffset), everything else scalar
  const unsigned plane_b = (unsigned)NT * (unsigned)KS * 1024u;      // bytes per W plane
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(Wf), 0, (int)(3u * plane_b), 0x00020000);
  const int voff = lane * 16;
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    const int nt0 = (pass * 4 + wave) * WTN;         // this wave's first 32-column subtile
    const bool active = nt0 < NT;                    // wave-uniform
    pf32x16 acc[2][WTN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B fragment of subtile j, plane q at k-step s: byte offset wb[j] + q * plane_b + s * 1024 (all scalar)
    unsigned wb[WTN];
#pragma unroll
    for (int j = 0; j < WTN; ++j) wb[j] = (unsigned)min(nt0 + j, NT - 1) * (unsigned)KS * 1024u;

    pbf16x8 a0[2][3], a1[2][3], b0[WTN][3], b1[WTN][3];
#define PN_LOAD_B(B, s)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < WTN; ++j) _Pragma("unroll") for (int q = 0; q < 3; ++q)                  \
      B[j][q] = __builtin_bit_cast(pbf16x8, __builtin_amdgcn_raw_buffer_load_b128(                               \
          wrsrc, voff, (int)(wb[j] + q * plane_b + (unsigned)(s) * 1024u), 0));
#define PN_LOAD_A(Afr, ksl)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 3; ++q)                    \
      Afr[i][q] = *reinterpret_cast<const pbf16x8 *>(&sA[q][i * 32 + fr][(ksl) * 16 + fk]);
#define PN_MMA(Afr, B)                                                                                           \
  if constexpr (DBG & 4) {                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                              \
      pn_keep(Afr[0][q]); pn_keep(Afr[1][q]);                                                                    \
      _Pragma("unroll") for (int j = 0; j < WTN; ++j) pn_keep(B[j][q]);                                          \
    }                                                                                                            \
  } else _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < WTN; ++j) {         \
    pf32x16 c = acc[i][j];                                                                                       \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[i][0], B[j][2], c, 0, 0, 0);                                 \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[i][2], B[j][0], c, 0, 0, 0);                                 \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[i][1], B[j][1], c, 0, 0, 0);                                 \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[i][0], B[j][1], c, 0, 0, 0);                                 \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[i][1], B[j][0], c, 0, 0, 0);                                 \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[i][0], B[j][0], c, 0, 0, 0);                                 \
    acc[i][j] = c;                                                                                               \
  }

    float bv[WTN];       // bias of this lane's column per subtile, loaded here so that the k loop covers its latency
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int col = min((nt0 + j) * 32 + fr, N - 1);
      bv[j] = bias ? bias[col] : 0.f;
    }
    if (active) {
      PN_LOAD_B(b0, 0)
      if constexpr (DBG & 1) PN_LOAD_B(b1, 0)
    }
#pragma unroll 1
    for (int kc = 0; kc < nchunks; ++kc) {
      if (!single) {
        __syncthreads();                 // every wave is done reading the previous chunk
        if constexpr (!(DBG & 8)) PN_STORE_CHUNK(kc)
        __syncthreads();
        // the next chunk this workgroup will stage (the first chunk of the next pass after the last one of this pass)
        const int nxt = kc + 1 < nchunks ? kc + 1 : 0;
        if constexpr (!(DBG & 8))
          if (kc + 1 < nchunks || pass + 1 < npass) PN_LOAD_CHUNK(nxt)
      }
      if (active) {
        const int k0 = kc * 8, ksteps = min(8, KS - k0);
        int ks = 0;
        PN_LOAD_A(a0, 0)
#pragma unroll 1
        for (; ks + 2 <= ksteps; ks += 2) {
          if constexpr (!(DBG & 1)) PN_LOAD_B(b1, k0 + ks + 1)
          if constexpr (!(DBG & 16)) PN_LOAD_A(a1, ks + 1) else if (ks == 0) PN_LOAD_A(a1, 1)
          PN_MMA(a0, b0)
          if constexpr (!(DBG & 1)) PN_LOAD_B(b0, min(k0 + ks + 2, KS - 1))   // the next chunk's first k-step when this chunk ends here
          if constexpr (!(DBG & 16)) PN_LOAD_A(a0, min(ks + 2, 7))
          PN_MMA(a1, b1)
        }
        if (ks < ksteps) {                              // odd tail: only ever the last chunk of K
          PN_MMA(a0, b0)
        }
      }
    }
#undef PN_LOAD_B
#undef PN_LOAD_A
#undef PN_MMA

    // ---- epilogue: straight from the accumulator layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5));
    //      whole panels / whole subtiles (the common case) store without any per-lane condition
    if (active) {
      const int h = lane >> 5;
      const bool rows_full = m0 + PN_BM <= M;            // workgroup-uniform
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        if (nt0 + j >= NT) break;                        // wave-uniform
        const int col = (nt0 + j) * 32 + fr;
        const bool cols_full = (nt0 + j) * 32 + 32 <= N; // wave-uniform
        // several Linears over the same rows as ONE launch (weights concatenated): output tensor o = col / out_cols
        // (wave-uniform: out_cols is a multiple of 32), column col - o * out_cols inside it
        const int otile = ((nt0 + j) * 32) / out_cols;
        const int64_t obase = (int64_t)otile * out_stride - (int64_t)otile * out_cols;
#define PN_EPI(GUARD)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int r = 0; r < 16; ++r) {                 \
    const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;                                                \
    float v = acc[i][j][r] + bv[j];                                                                              \
    if (ACT == 1) v = gelu_exact(v);                                                                             \
    if ((DBG & 2) ? (row < 0) : (GUARD)) {                                                                       \
      if constexpr (D16) {                                                                                       \
        reinterpret_cast<_Float16 *>(Dv)[obase + (int64_t)row * ldd + col] = (_Float16)v;                        \
      } else {                                                                                                   \
        float *dst = reinterpret_cast<float *>(Dv) + obase + (int64_t)row * ldd + col;                           \
        if (nt_store) __builtin_nontemporal_store(v, dst);                                                       \
        else *dst = v;                                                                                           \
      }                                                                                                          \
    }                                                                                                            \
  }
        if (rows_full && cols_full) PN_EPI(true)
        else PN_EPI(col < N && row < M)
#undef PN_EPI
      }
    }
  }
#undef PN_LOAD_CHUNK
#undef PN_STORE_CHUNK
}

}  // namespace isg

using namespace isg;

static inline int pn_ksteps(int K) { return (K + 15) / 16; }
static inline int pn_subtiles(int N) { return (N + 31) / 32; }

extern "C" int64_t isg_split_bf16x3_frag_elems(int64_t rows, int32_t K) {
  if (rows <= 0 || K <= 0) return 0;
  return 3ll * pn_subtiles((int)rows) * pn_ksteps(K) * 512;
}

extern "C" int isg_split_bf16x3_frag(const float *w, int64_t rows, int32_t K, uint16_t *planes, void *stream) {
  if (rows < 0 || K <= 0) return ISG_EINVAL;
  if (rows == 0) return ISG_OK;
  if (!w || !planes) return ISG_EINVAL;
  if (rows >= (1ll << 24)) return ISG_EUNSUPPORTED;
  const int NT = pn_subtiles((int)rows), KS = pn_ksteps(K);
  const int64_t total = (int64_t)NT * KS * 512;
  if ((total + 255) / 256 >= (1ll << 31)) return ISG_EUNSUPPORTED;
  split_bf16x3_frag_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(
      w, (int)rows, K, NT, KS, reinterpret_cast<__bf16 *>(planes));
  return check_launch();
}

// results at least this large are written with non-temporal stores (see isg_gemm.hip: the big projected rows are
// consumed by a LATER kernel and should not displace the operands); read once, at load
static long long pn_nt_bytes() {
  static const long long v = [] {
    const char *e = getenv("ISG_GEMM_NT_MB");
    return (e ? atoll(e) : 128) * 1000000ll;
  }();
  return v;
}

static int panel_launch(const void *a, int32_t a_is_f16, const uint16_t *w_frag, const float *bias, void *d,
                        int32_t d_is_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                        int32_t out_cols, int64_t out_stride, void *stream) {
  if (M < 0 || N <= 0 || K <= 0 || lda < K || ldd < out_cols || act < 0 || act > 1) return ISG_EINVAL;
  if (out_cols <= 0 || (out_cols < N && (out_cols & 31) != 0) || N % out_cols != 0) return ISG_EINVAL;
  if (M == 0) return ISG_OK;
  if (!a || !w_frag || !d) return ISG_EINVAL;
  const bool a16 = a_is_f16 != 0, d16 = d_is_f16 != 0;
  const uintptr_t amask = a16 ? 7 : 15;
  if ((K & 3) != 0 || (lda & 3) != 0 || (reinterpret_cast<uintptr_t>(a) & amask) != 0) return ISG_EUNSUPPORTED;
  const long long panels = (M + PN_BM - 1) / PN_BM;
  if (panels >= (1ll << 31) || M >= (1ll << 31)) return ISG_EUNSUPPORTED;
  const int NT = pn_subtiles(N), KS = pn_ksteps(K);
  if (3ll * NT * KS * 1024 >= (1ll << 31)) return ISG_EUNSUPPORTED;   // the W planes are addressed with 32-bit offsets
  // subtiles per wave and pass: the wider tile unless it leaves much more of the last pass idle
  auto waste = [&](int w) { const int per = 4 * w; return ((NT + per - 1) / per) * per - NT; };
  int wtn = waste(2) <= waste(1) + 1 ? 2 : 1;
  {
    static const int force = [] { const char *e = getenv("ISG_PANEL_WTN"); return e ? atoi(e) : 0; }();
    if (force == 1 || force == 2) wtn = force;
  }
  const long long nt_b = pn_nt_bytes();
  const int nt = nt_b >= 0 && (long long)M * N * (d16 ? 2 : 4) >= nt_b;
  const __bf16 *wf = reinterpret_cast<const __bf16 *>(w_frag);
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)panels), block(PN_THREADS);
#ifdef ISG_PANEL_ABLATION   // profiling build only (tools/ablate_panel.py): an ablated variant selected by the environment
  {
    const char *dv = getenv("ISG_PANEL_DBG");
    const int dbg = dv ? atoi(dv) : 0;
#define ISG_PN_DBG(v)                                                                                                   \
  if (dbg == v && !a16 && !d16) {                                                                                       \
    if (KS <= 8) linear_panel_kernel<0, 2, false, false, true, v><<<grid, block, 0, st>>>(a, wf, bias, d, (int)M, N, K, KS, NT, lda, ldd, nt); \
    else linear_panel_kernel<0, 2, false, false, false, v><<<grid, block, 0, st>>>(a, wf, bias, d, (int)M, N, K, KS, NT, lda, ldd, nt, out_cols, (long long)out_stride); \
    return check_launch();                                                                                              \
  }
    ISG_PN_DBG(1) ISG_PN_DBG(2) ISG_PN_DBG(3) ISG_PN_DBG(4) ISG_PN_DBG(8) ISG_PN_DBG(16) ISG_PN_DBG(17) ISG_PN_DBG(6)
    ISG_PN_DBG(7) ISG_PN_DBG(10) ISG_PN_DBG(27) ISG_PN_DBG(31)
#undef ISG_PN_DBG
  }
#endif
#define ISG_PN(ACT_, W_, A_, D_) \
  do {                                                                                                             \
    if (KS <= 8) linear_panel_kernel<ACT_, W_, A_, D_, true><<<grid, block, 0, st>>>(a, wf, bias, d, (int)M, N, K, KS, NT, lda, ldd, nt, out_cols, (long long)out_stride); \
    else linear_panel_kernel<ACT_, W_, A_, D_, false><<<grid, block, 0, st>>>(a, wf, bias, d, (int)M, N, K, KS, NT, lda, ldd, nt, out_cols, (long long)out_stride); \
  } while (0)
#define ISG_PN_AD(ACT_, W_)                                  \
  do {                                                       \
    if (!a16 && !d16) ISG_PN(ACT_, W_, false, false);        \
    else if (a16 && !d16) ISG_PN(ACT_, W_, true, false);     \
    else if (!a16 && d16) ISG_PN(ACT_, W_, false, true);     \
    else ISG_PN(ACT_, W_, true, true);                       \
  } while (0)
  if (act == 1) { if (wtn == 2) ISG_PN_AD(1, 2); else ISG_PN_AD(1, 1); }
  else { if (wtn == 2) ISG_PN_AD(0, 2); else ISG_PN_AD(0, 1); }
#undef ISG_PN_AD
#undef ISG_PN
  return check_launch();
}

extern "C" int isg_linear_panel(const void *a, int32_t a_is_f16, const uint16_t *w_frag, const float *bias, void *d,
                                int32_t d_is_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                                void *stream) {
  return panel_launch(a, a_is_f16, w_frag, bias, d, d_is_f16, M, N, K, lda, ldd, act, N, 0, stream);
}

extern "C" int isg_linear_panel_multi(const void *a, int32_t a_is_f16, const uint16_t *w_frag, const float *bias, void *d,
                                      int32_t d_is_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd,
                                      int32_t act, int32_t out_cols, int64_t out_stride, void *stream) {
  return panel_launch(a, a_is_f16, w_frag, bias, d, d_is_f16, M, N, K, lda, ldd, act, out_cols, out_stride, stream);
}
