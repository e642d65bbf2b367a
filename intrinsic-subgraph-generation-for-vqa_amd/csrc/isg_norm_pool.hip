// Per-graph kernels of the MGAT layer tail and the read-out: instruction->node scatter-softmax
// attention, GraphNorm, residual, and question-conditioned attention pooling.
//
// One workgroup owns one graph (its nodes are a contiguous row range [ptr[g], ptr[g+1])).
//   phase A  waves stride over the graph's nodes; a wave computes one <query, key_n> dot product
//            with a 64-lane DPP/shuffle butterfly and parks the scaled logit in LDS
//   phase B  wave 0 turns the LDS strip into softmax weights (max, exp, ordered sum, divide)
//   phase C  threads stride over channels; each walks the graph's nodes in order, so per-graph
//            means / variances / pooled sums accumulate in node order with unfused mul+add --
//            the order and roundings of the reference's CPU scatter kernels
// Rows are re-read from L1/L2 in phase C (a 20-node x 128-channel graph is 10 KB).  A graph is ONE workgroup, so a large graph
// bounds the launch (BASELINE configs[4]'s 200-node graphs: 155 us for 67 MB of traffic with two waves taking one node each per
// round trip): phase A keeps four nodes per wave in flight on at least four waves, the elementwise last pass of the layer tail
// runs on every thread.  (Keeping the graph's first 16 or 32 KB of rows in LDS on top of that bought nothing on configs[4] and
// cost the C = 300 model 0.06-0.08 ms per step in occupancy: profiles/r04_au_per_graph_tail.txt.)
#include <algorithm>

#include "isg_common.hpp"

namespace isg {

constexpr int GN_NCAP = 1024;  // nodes per graph the LDS strip holds (host rejects larger graphs)
constexpr int GN_CMAX = 512;    // channels whose per-graph statistics fit the LDS strips of the layer tail's last pass

#ifndef ISG_PL_U
#define ISG_PL_U 4
#endif
#ifndef ISG_GT_U
#define ISG_GT_U 16
#endif
constexpr int PL_U = ISG_PL_U;   // nodes a wave has in flight in phase A
constexpr int GT_U = ISG_GT_U;   // nodes a thread has in flight in the node-ordered passes of phase C

// phase A: s_a[k] = <q, key[nb+k] (* mask)> / denom.  A wave takes PL_U nodes per round: their loads are in flight together (one
// node per round was one exposed memory round trip per node and wave -- 100 of them in a row for a 200-node graph on two waves).
// A lane's partial sum still runs over its float4 columns in ascending order and the 64-lane butterfly is the same: same bits.
__device__ __forceinline__ void phase_logits(const float4 *__restrict__ q4, const float4 *__restrict__ key4,
                                             const float *__restrict__ node_mask, int nb, int n, int Q, float denom,
                                             float *s_a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int k0 = PL_U * wave; k0 < n; k0 += PL_U * nw) {
    float part[PL_U], m[PL_U];
#pragma unroll
    for (int u = 0; u < PL_U; ++u) {
      part[u] = 0.f;
      m[u] = node_mask ? node_mask[nb + min(k0 + u, n - 1)] : 1.f;
    }
    for (int c = lane; c < Q; c += 64) {
      float4 v[PL_U];
#pragma unroll
      for (int u = 0; u < PL_U; ++u) v[u] = key4[(size_t)(nb + min(k0 + u, n - 1)) * Q + c];
      float4 qv = q4[c];        // scalars of their own: the products of PL_U rows pair up across rows, never (q.y, q.x) out of one pair
      qv.x = own_reg(qv.x); qv.y = own_reg(qv.y); qv.z = own_reg(qv.z); qv.w = own_reg(qv.w);
#pragma unroll
      for (int u = 0; u < PL_U; ++u) {
        if (node_mask) {
          v[u].x = mul_rn(v[u].x, m[u]); v[u].y = mul_rn(v[u].y, m[u]);
          v[u].z = mul_rn(v[u].z, m[u]); v[u].w = mul_rn(v[u].w, m[u]);
        }
        part[u] += dot4_rn(v[u], qv);
      }
    }
#pragma unroll
    for (int u = 0; u < PL_U; ++u) {
      const float dot = wave_sum(part[u]);
      if (lane == 0 && k0 + u < n) s_a[k0 + u] = dot / denom;
    }
  }
}

// phase B (wave 0 only): s_a[k] <- exp(s_a[k]-max) / (sum + eps_add); the sum runs in node order.
__device__ __forceinline__ void phase_softmax(int n, float eps_add, float *s_a) {
  const int lane = threadIdx.x & 63;
  float mx = -INFINITY;
  for (int k = lane; k < n; k += 64) mx = fmaxf(mx, s_a[k]);
  mx = wave_max(mx);
  for (int k = lane; k < n; k += 64) s_a[k] = expf(s_a[k] - mx);
  __builtin_amdgcn_wave_barrier();
  float sum = 0.f;
  if (n <= 128) {
    for (int k = 0; k < n; ++k) sum += s_a[k];   // every lane walks the strip: LDS broadcast reads
  } else {
    const int per = (n + 63) / 64;
    float part = 0.f;
    for (int k = lane * per; k < min(n, (lane + 1) * per); ++k) part += s_a[k];
    sum = wave_sum(part);
  }
  sum += eps_add;
  __builtin_amdgcn_wave_barrier();
  for (int k = lane; k < n; k += 64) s_a[k] = s_a[k] / sum;
}

// MODE 0: out = a_n * value_n                      (scatter_scaled_dot_product_attention)
// MODE 1: out = GraphNorm(x)                        (x = key)
// MODE 2: out = GraphNorm(a_n * value_n) + h  [* node_mask]   (fused MGAT layer tail)
#ifndef ISG_TAIL_UNROLL
#define ISG_TAIL_UNROLL 8
#endif
template <int MODE>
__global__ __launch_bounds__(512) void graph_tail_kernel(const float *__restrict__ query, const float *__restrict__ key,
                                                         const float *h, const int *__restrict__ ptr,
                                                         const float *__restrict__ weight, const float *__restrict__ bias,
                                                         const float *__restrict__ mean_scale, float eps,
                                                         const float *__restrict__ node_mask, float *out,
                                                         int C, float denom) {
  __shared__ float s_a[GN_NCAP];
  __shared__ float s_mean[MODE == 0 ? 1 : GN_CMAX], s_std[MODE == 0 ? 1 : GN_CMAX];
  const int g = blockIdx.x;
  const int nb = ptr[g];
  const int n = MODE == 1 ? ptr[g + 1] - nb : min(ptr[g + 1] - nb, GN_NCAP);
  if (n <= 0) return;
  if (MODE != 1) {
    phase_logits((const float4 *)(query + (size_t)g * C), (const float4 *)key, nullptr, nb, n, C >> 2, denom, s_a);
    __syncthreads();
    if (threadIdx.x < 64) phase_softmax(n, 0.f, s_a);
    __syncthreads();
  }
  if (MODE == 0) {   // h carries `value` here
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
      const float *val = h + (size_t)nb * C + ch;
      for (int k = 0; k < n; ++k) out[(size_t)(nb + k) * C + ch] = mul_rn(s_a[k], val[(size_t)k * C]);
    }
    return;
  }
  const float cnt = (float)n;
  const bool strips = C <= GN_CMAX && (int)blockDim.x >= C;     // the statistics go through LDS, the last pass runs on every thread
#define GT_VALUE(k, src) (MODE == 2 ? mul_rn(s_a[k], (src)) : (src))
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const float *col = key + (size_t)nb * C + ch;
    // blocks of GT_U nodes with every load of a block in flight, the last block's too (indices clamped, sums predicated): an
    // unrolled loop's remainder ran one exposed load per node
    float sum = 0.f;
    for (int k0 = 0; k0 < n; k0 += GT_U) {
      float v[GT_U];
#pragma unroll
      for (int u = 0; u < GT_U; ++u) v[u] = col[(size_t)min(k0 + u, n - 1) * C];
#pragma unroll
      for (int u = 0; u < GT_U; ++u) {
        const float t = add_rn(sum, GT_VALUE(min(k0 + u, n - 1), v[u]));
        sum = k0 + u < n ? t : sum;
      }
    }
    const float mean_ms = mul_rn(sum / cnt, mean_scale[ch]);
    float sq = 0.f;
    for (int k0 = 0; k0 < n; k0 += GT_U) {
      float v[GT_U];
#pragma unroll
      for (int u = 0; u < GT_U; ++u) v[u] = col[(size_t)min(k0 + u, n - 1) * C];
#pragma unroll
      for (int u = 0; u < GT_U; ++u) {
        const float o = sub_rn(GT_VALUE(min(k0 + u, n - 1), v[u]), mean_ms);
        const float t = add_rn(sq, mul_rn(o, o));
        sq = k0 + u < n ? t : sq;
      }
    }
    const float stdv = sqrtf(add_rn(sq / cnt, eps));
    if (strips) {
      s_mean[ch] = mean_ms;
      s_std[ch] = stdv;
      continue;
    }
    const float w = weight[ch], b = bias[ch];
#pragma unroll ISG_TAIL_UNROLL
    for (int k = 0; k < n; ++k) {
      const float o = sub_rn(GT_VALUE(k, col[(size_t)k * C]), mean_ms);
      float y = add_rn(mul_rn(w, o) / stdv, b);
      if (MODE == 2) {
        y = add_rn(y, h[(size_t)(nb + k) * C + ch]);
        if (node_mask) y = mul_rn(node_mask[nb + k], y);
      }
      out[(size_t)(nb + k) * C + ch] = y;
    }
  }
  if (!strips) return;
  __syncthreads();
  // the last pass has no order to keep: thread = (row lane, channel), the workgroup's spare threads take every other row
  const int nrl = (int)blockDim.x / C, rl = (int)threadIdx.x / C, ch = (int)threadIdx.x - rl * C;
  if (rl >= nrl) return;
  const float *col = key + (size_t)nb * C + ch;
  const float mean_ms = s_mean[ch], stdv = s_std[ch], w = weight[ch], b = bias[ch];
  for (int k0 = rl; k0 < n; k0 += nrl * GT_U) {
    float v[GT_U], hv[GT_U];
#pragma unroll
    for (int u = 0; u < GT_U; ++u) {
      const int k = min(k0 + u * nrl, n - 1);
      v[u] = col[(size_t)k * C];
      hv[u] = MODE == 2 ? h[(size_t)(nb + k) * C + ch] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < GT_U; ++u) {
      const int k = k0 + u * nrl;
      if (k >= n) break;
      const float o = sub_rn(GT_VALUE(k, v[u]), mean_ms);
      float y = add_rn(mul_rn(w, o) / stdv, b);
      if (MODE == 2) {
        y = add_rn(y, hv[u]);
        if (node_mask) y = mul_rn(node_mask[nb + k], y);
      }
      out[(size_t)(nb + k) * C + ch] = y;
    }
  }
#undef GT_VALUE
}

// GraphNorm with every intermediate in double (scene_graph_encoder.py:99-102).
__global__ __launch_bounds__(512) void graph_norm_f64_kernel(const float *__restrict__ x, const int *__restrict__ ptr,
                                                             const float *__restrict__ weight, const float *__restrict__ bias,
                                                             const float *__restrict__ mean_scale, double eps,
                                                             float *__restrict__ out, int C) {
  const int g = blockIdx.x;
  const int nb = ptr[g];
  const int n = ptr[g + 1] - nb;
  if (n <= 0) return;
  const double cnt = (double)n;
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const float *col = x + (size_t)nb * C + ch;
    double sum = 0.0;
    for (int k = 0; k < n; ++k) sum = dadd_rn(sum, (double)col[(size_t)k * C]);
    const double mean_ms = dmul_rn(sum / cnt, (double)mean_scale[ch]);
    double sq = 0.0;
    for (int k = 0; k < n; ++k) {
      const double o = dsub_rn((double)col[(size_t)k * C], mean_ms);
      sq = dadd_rn(sq, dmul_rn(o, o));
    }
    const double stdv = sqrt(dadd_rn(sq / cnt, eps));
    const double w = (double)weight[ch], b = (double)bias[ch];
    for (int k = 0; k < n; ++k) {
      const double o = dsub_rn((double)col[(size_t)k * C], mean_ms);
      out[(size_t)(nb + k) * C + ch] = (float)dadd_rn(dmul_rn(w, o) / stdv, b);
    }
  }
}

__global__ __launch_bounds__(512) void global_attn_pool_kernel(const float *__restrict__ xn, const float *__restrict__ q,
                                                               const int *__restrict__ ptr, const float *__restrict__ node_mask,
                                                               float *__restrict__ out, float *__restrict__ gate, int C,
                                                               float denom) {
  __shared__ float s_a[GN_NCAP];
  const int g = blockIdx.x;
  const int nb = ptr[g];
  const int n = min(ptr[g + 1] - nb, GN_NCAP);
  if (n <= 0) {   // scatter_add leaves an empty graph's row at zero
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) out[(size_t)g * C + ch] = 0.f;
    return;
  }
  phase_logits((const float4 *)(q + (size_t)g * C), (const float4 *)xn, node_mask, nb, n, C >> 2, denom, s_a);
  __syncthreads();
  if (threadIdx.x < 64) phase_softmax(n, 1e-16f, s_a);
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += blockDim.x) gate[nb + k] = s_a[k];
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const float *col = xn + (size_t)nb * C + ch;
    float sum = 0.f;
    for (int k0 = 0; k0 < n; k0 += GT_U) {
      float v[GT_U];
#pragma unroll
      for (int u = 0; u < GT_U; ++u) v[u] = col[(size_t)min(k0 + u, n - 1) * C];
#pragma unroll
      for (int u = 0; u < GT_U; ++u) {
        const int k = min(k0 + u, n - 1);
        if (node_mask) v[u] = mul_rn(v[u], node_mask[nb + k]);
        const float t = add_rn(sum, mul_rn(s_a[k], v[u]));
        sum = k0 + u < n ? t : sum;
      }
    }
    out[(size_t)g * C + ch] = sum;
  }
}

// threads per graph: one per channel in phase C (a channel count just above a block size -- the reference's C = 300 on 256 threads --
// made a second, mostly idle round of three passes over the graph's column)
static int block_for(int C) { return C <= 64 ? 64 : (C <= 128 ? 128 : (C <= 256 ? 256 : (C <= 320 ? 320 : (C <= 384 ? 384 : 512)))); }

}  // namespace isg

using namespace isg;

static int check_bc(int64_t B, int32_t C) {
  if (B < 0 || C <= 0) return ISG_EINVAL;
  if ((C & 3) != 0 || B >= (1ll << 31)) return ISG_EUNSUPPORTED;
  return ISG_OK;
}

extern "C" int isg_scatter_attention(const float *query, const float *key, const float *value, const int32_t *ptr,
                                     float *out, int64_t B, int32_t C, void *stream) {
  int st = check_bc(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!query || !key || !value || !ptr || !out) return ISG_EINVAL;
  graph_tail_kernel<0><<<(unsigned)B, std::max(256, block_for(C)), 0, as_stream(stream)>>>(
      query, key, value, ptr, nullptr, nullptr, nullptr, 0.f, nullptr, out, C, (float)sqrt((double)C));
  return check_launch();
}

extern "C" int isg_graph_norm(const float *x, const int32_t *ptr, const float *weight, const float *bias,
                              const float *mean_scale, double eps, int32_t accumulate_fp64, float *out, int64_t B,
                              int32_t C, void *stream) {
  int st = check_bc(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!x || !ptr || !weight || !bias || !mean_scale || !out) return ISG_EINVAL;
  if (accumulate_fp64)
    graph_norm_f64_kernel<<<(unsigned)B, block_for(C), 0, as_stream(stream)>>>(x, ptr, weight, bias, mean_scale, eps, out, C);
  else
    graph_tail_kernel<1><<<(unsigned)B, block_for(C), 0, as_stream(stream)>>>(
        nullptr, x, nullptr, ptr, weight, bias, mean_scale, (float)eps, nullptr, out, C, 1.f);
  return check_launch();
}

extern "C" int isg_instr_attn_graphnorm_residual(const float *ins, const float *c, const float *h, const int32_t *ptr,
                                                 const float *weight, const float *bias, const float *mean_scale,
                                                 double eps, const float *node_mask, float *h_out, int64_t B, int32_t C,
                                                 void *stream) {
  int st = check_bc(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!ins || !c || !h || !ptr || !weight || !bias || !mean_scale || !h_out) return ISG_EINVAL;
  // scatter_scaled_dot_product.py:11 divides by math.sqrt(C): a double rounded to fp32 by the tensor op
  // at least four waves: phase A keeps 16 of the graph's rows in flight, the last pass spreads the rows over the spare threads
  graph_tail_kernel<2><<<(unsigned)B, std::max(256, block_for(C)), 0, as_stream(stream)>>>(
      ins, c, h, ptr, weight, bias, mean_scale, (float)eps, node_mask, h_out, C, (float)sqrt((double)C));
  return check_launch();
}

extern "C" int isg_global_attn_pool(const float *xn, const float *q, const int32_t *ptr, const float *node_mask,
                                    float *out, float *gate, int64_t B, int32_t C, void *stream) {
  int st = check_bc(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!xn || !q || !ptr || !out || !gate) return ISG_EINVAL;
  // att_pooling.py:68 divides by torch.sqrt(torch.tensor(C)): fp32 sqrt
  global_attn_pool_kernel<<<(unsigned)B, std::max(256, block_for(C)), 0, as_stream(stream)>>>(xn, q, ptr, node_mask, out, gate, C,
                                                                              sqrtf((float)C));
  return check_launch();
}
