// Backward of the per-graph operators around the message passing: the fused MGAT layer tail
// (instruction attention -> GraphNorm -> residual [-> mask]), the question-conditioned pooling, the instruction gate and
// the node gate.  The reference differentiates these with plain autograd over PyG / torch_scatter ops
// (mgat.py:168-177, att_pooling.py:63-73, mgat_v2_conv.py:156-157, masking.py:151-155).
//
// One workgroup owns one graph, like the forward kernels (isg_norm_pool.hip): softmax weights and per-node scalars in an
// LDS strip, per-channel statistics in LDS, channel-parallel column walks in node order, wave-per-node row reductions.
// Parameter gradients that sum over ALL graphs (GraphNorm weight / bias / mean_scale, rows of q shared by several
// graphs) are written as one partial row per graph and summed by the caller: no atomics, bitwise reproducible.
#include "isg_common.hpp"

namespace isg {

constexpr int TB_NCAP = 1024;   // nodes per graph (same bound as the forward strip)
constexpr int TB_CCAP = 1024;   // channels

__device__ __forceinline__ float gelu_grad(float t) {   // d/dt [0.5 t (1 + erf(t / sqrt 2))]
  return 0.5f * (1.0f + erff(t * 0.70710678118654752440f)) + t * 0.3989422804014327f * expf(-0.5f * t * t);
}

// softmax over s[0..n) in place (wave 0), denominator + eps_add; same arithmetic as the forward's phase_softmax
__device__ __forceinline__ void tb_softmax(int n, float eps_add, float *s) {
  const int lane = threadIdx.x & 63;
  float mx = -INFINITY;
  for (int k = lane; k < n; k += 64) mx = fmaxf(mx, s[k]);
  mx = wave_max(mx);
  for (int k = lane; k < n; k += 64) s[k] = expf(s[k] - mx);
  __builtin_amdgcn_wave_barrier();
  float part = 0.f;
  for (int k = lane; k < n; k += 64) part += s[k];
  const float sum = wave_sum(part) + eps_add;
  __builtin_amdgcn_wave_barrier();
  for (int k = lane; k < n; k += 64) s[k] = s[k] / sum;
}

// ---- layer tail ------------------------------------------------------------------------------------------------------
// forward: a = softmax_g(<ins, c_k> / sqrt C);  v_k = a_k c_k;  o = v - mean(v) * ms;  y = w o / sqrt(var(o) + eps) + b;
//          out_k = (y_k + h_k) [* m_k]
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float *__restrict__ ins, const float *__restrict__ c,
                                                       const float *__restrict__ h, const int *__restrict__ ptr,
                                                       const float *__restrict__ weight, const float *__restrict__ bias,
                                                       const float *__restrict__ mean_scale, float eps,
                                                       const float *__restrict__ node_mask, const float *__restrict__ g_out,
                                                       float *__restrict__ d_ins, float *__restrict__ d_c,
                                                       float *__restrict__ d_h, float *__restrict__ d_mask,
                                                       float *__restrict__ partial, int C, float denom) {
  __shared__ float s_a[TB_NCAP], s_t[TB_NCAP];      // softmax weights; d a_k, then d logit_k
  __shared__ float s_mean[TB_CCAP], s_rstd[TB_CCAP];
  const int g = blockIdx.x;
  const int nb = ptr[g], n = min(ptr[g + 1] - nb, TB_NCAP);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float *pw = partial + (size_t)g * 3 * C;          // [d weight | d bias | d mean_scale] of this graph
  if (n <= 0) {
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
      d_ins[(size_t)g * C + ch] = 0.f;
      pw[ch] = pw[C + ch] = pw[2 * C + ch] = 0.f;
    }
    return;
  }
  const float *q = ins + (size_t)g * C;
  // A: attention weights, as in the forward
  for (int k = wave; k < n; k += nw) {
    const float *row = c + (size_t)(nb + k) * C;
    float part = 0.f;
    for (int ch = lane; ch < C; ch += 64) part += row[ch] * q[ch];
    const float dot = wave_sum(part);
    if (lane == 0) s_a[k] = dot / denom;
  }
  __syncthreads();
  if (threadIdx.x < 64) tb_softmax(n, 0.f, s_a);
  __syncthreads();
  // B: per channel, GraphNorm backward down to d v; d v is parked in d_c, d h written
  const float cnt = (float)n;
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const float *col = c + (size_t)nb * C + ch;
    const float *gcol = g_out + (size_t)nb * C + ch;
    float sum = 0.f;
    for (int k = 0; k < n; ++k) sum += s_a[k] * col[(size_t)k * C];
    const float mean = sum / cnt, ms = mean_scale[ch], mean_ms = mean * ms;
    float sq = 0.f;
    for (int k = 0; k < n; ++k) {
      const float o = s_a[k] * col[(size_t)k * C] - mean_ms;
      sq += o * o;
    }
    const float var = sq / cnt, rstd = 1.0f / sqrtf(var + eps), w = weight[ch];
    s_mean[ch] = mean_ms;
    s_rstd[ch] = rstd;
    float db = 0.f, dw = 0.f, gwo = 0.f;
    for (int k = 0; k < n; ++k) {
      const float gk = gcol[(size_t)k * C] * (node_mask ? node_mask[nb + k] : 1.f);
      const float o = s_a[k] * col[(size_t)k * C] - mean_ms;
      db += gk;
      dw += gk * o * rstd;
      gwo += gk * w * o;
      d_h[(size_t)(nb + k) * C + ch] = gk;
    }
    const float dvar = -0.5f * gwo * rstd * rstd * rstd;
    float sdo = 0.f;
    for (int k = 0; k < n; ++k) {
      const float gk = gcol[(size_t)k * C] * (node_mask ? node_mask[nb + k] : 1.f);
      const float o = s_a[k] * col[(size_t)k * C] - mean_ms;
      sdo += gk * w * rstd + dvar * 2.0f * o / cnt;
    }
    for (int k = 0; k < n; ++k) {
      const float gk = gcol[(size_t)k * C] * (node_mask ? node_mask[nb + k] : 1.f);
      const float o = s_a[k] * col[(size_t)k * C] - mean_ms;
      const float d_o = gk * w * rstd + dvar * 2.0f * o / cnt;
      d_c[(size_t)(nb + k) * C + ch] = d_o - ms * sdo / cnt;       // d v_k (finalised in D)
    }
    pw[ch] = dw;
    pw[C + ch] = db;
    pw[2 * C + ch] = -mean * sdo;
  }
  __syncthreads();
  // C: per node, d a_k = <d v_k, c_k>  and  d m_k = <g_k, y_k + h_k>
  for (int k = wave; k < n; k += nw) {
    const float *crow = c + (size_t)(nb + k) * C, *dv = d_c + (size_t)(nb + k) * C;
    const float ak = s_a[k];
    float pa = 0.f, pm = 0.f;
    for (int ch = lane; ch < C; ch += 64) {
      pa += dv[ch] * crow[ch];
      if (d_mask) {
        const float y = weight[ch] * (ak * crow[ch] - s_mean[ch]) * s_rstd[ch] + bias[ch];
        pm += g_out[(size_t)(nb + k) * C + ch] * (y + h[(size_t)(nb + k) * C + ch]);
      }
    }
    pa = wave_sum(pa);
    if (d_mask) pm = wave_sum(pm);
    if (lane == 0) {
      s_t[k] = pa;
      if (d_mask) d_mask[nb + k] = pm;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // softmax backward: d l_k = a_k (d a_k - sum_m a_m d a_m)
    float part = 0.f;
    for (int k = lane; k < n; k += 64) part += s_a[k] * s_t[k];
    const float dot = wave_sum(part);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < n; k += 64) s_t[k] = s_a[k] * (s_t[k] - dot);
  }
  __syncthreads();
  // D: d c_k = a_k d v_k + d l_k ins / sqrt C;  d ins = sum_k d l_k c_k / sqrt C
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const float qc = q[ch] / denom;
    float di = 0.f;
    for (int k = 0; k < n; ++k) {
      const size_t at = (size_t)(nb + k) * C + ch;
      di += s_t[k] * c[at];
      d_c[at] = s_a[k] * d_c[at] + s_t[k] * qc;
    }
    d_ins[(size_t)g * C + ch] = di / denom;
  }
}

// ---- pooling -----------------------------------------------------------------------------------------------------------
// forward: x_k = xn_k m_k;  gate = softmax_g(<x_k, q> / sqrt C) (+1e-16);  out = sum_k gate_k x_k
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float *__restrict__ xn, const float *__restrict__ q,
                                                       const int *__restrict__ ptr, const float *__restrict__ node_mask,
                                                       const float *__restrict__ g_out, const float *__restrict__ g_gate,
                                                       float *__restrict__ d_xn, float *__restrict__ d_q,
                                                       float *__restrict__ d_mask, int C, float denom) {
  __shared__ float s_a[TB_NCAP], s_t[TB_NCAP];
  const int g = blockIdx.x;
  const int nb = ptr[g], n = min(ptr[g + 1] - nb, TB_NCAP);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (n <= 0) {
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) d_q[(size_t)g * C + ch] = 0.f;
    return;
  }
  const float *qr = q + (size_t)g * C, *go = g_out + (size_t)g * C;
  for (int k = wave; k < n; k += nw) {   // logits and d gate_k = <d out, x_k> (+ upstream gate gradient)
    const float *row = xn + (size_t)(nb + k) * C;
    const float m = node_mask ? node_mask[nb + k] : 1.f;
    float pl = 0.f, pg = 0.f;
    for (int ch = lane; ch < C; ch += 64) {
      const float x = row[ch] * m;
      pl += x * qr[ch];
      pg += x * go[ch];
    }
    pl = wave_sum(pl);
    pg = wave_sum(pg);
    if (lane == 0) {
      s_a[k] = pl / denom;
      s_t[k] = pg + (g_gate ? g_gate[nb + k] : 0.f);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    tb_softmax(n, 1e-16f, s_a);
    __builtin_amdgcn_wave_barrier();
    float part = 0.f;
    for (int k = lane; k < n; k += 64) part += s_a[k] * s_t[k];
    const float dot = wave_sum(part);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < n; k += 64) s_t[k] = s_a[k] * (s_t[k] - dot);     // d logit_k
  }
  __syncthreads();
  for (int k = wave; k < n; k += nw) {   // d x_k = gate_k d out + d l_k q / sqrt C;  d xn = d x m;  d m = <d x, xn>
    const float *row = xn + (size_t)(nb + k) * C;
    float *drow = d_xn + (size_t)(nb + k) * C;
    const float m = node_mask ? node_mask[nb + k] : 1.f, ak = s_a[k], dl = s_t[k] / denom;
    float pm = 0.f;
    for (int ch = lane; ch < C; ch += 64) {
      const float dx = ak * go[ch] + dl * qr[ch];
      drow[ch] = dx * m;
      pm += dx * row[ch];
    }
    if (d_mask) {
      pm = wave_sum(pm);
      if (lane == 0) d_mask[nb + k] = pm;
    }
  }
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {   // d q = sum_k d l_k x_k / sqrt C
    float acc = 0.f;
    for (int k = 0; k < n; ++k)
      acc += s_t[k] * xn[(size_t)(nb + k) * C + ch] * (node_mask ? node_mask[nb + k] : 1.f);
    d_q[(size_t)g * C + ch] = acc / denom;
  }
}

// ---- instruction gate: y = gelu(x * instr[g]) ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void instr_gate_bwd_kernel(const float *__restrict__ x, const float *__restrict__ instr,
                                                             const int *__restrict__ ptr, const float *__restrict__ g_out,
                                                             float *__restrict__ d_x, float *__restrict__ d_instr, int C) {
  const int g = blockIdx.x;
  const int nb = ptr[g], n = ptr[g + 1] - nb;
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const float iv = instr[(size_t)g * C + ch];
    float acc = 0.f;
    for (int k = 0; k < n; ++k) {
      const size_t at = (size_t)(nb + k) * C + ch;
      const float xv = x[at];
      const float dt = g_out[at] * gelu_grad(xv * iv);
      d_x[at] = dt * iv;
      acc += dt * xv;
    }
    d_instr[(size_t)g * C + ch] = acc;
  }
}

// ---- node gate: gate_n = gelu(<xn_n, q[r]> / sqrt C), r = batch[g] (double index) or g ------------------------------------
__global__ __launch_bounds__(256) void node_gate_bwd_kernel(const float *__restrict__ xn, const float *__restrict__ q,
                                                            const int64_t *__restrict__ batch, int dbl,
                                                            const int *__restrict__ ptr, const float *__restrict__ g_out,
                                                            float *__restrict__ d_xn, float *__restrict__ d_q_part,
                                                            int N, int C, float denom) {
  __shared__ float s_ds[TB_NCAP];
  const int g = blockIdx.x;
  const int nb = ptr[g], n = min(ptr[g + 1] - nb, TB_NCAP);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (n <= 0) {
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) d_q_part[(size_t)g * C + ch] = 0.f;
    return;
  }
  // every node of graph g reads the same row: batch[batch[n]] = batch[g] (quirk Q3), or g itself
  const int64_t r = dbl ? batch[min((int64_t)g, (int64_t)N - 1)] : (int64_t)g;
  const float *qr = q + (size_t)r * C;
  for (int k = wave; k < n; k += nw) {
    const float *row = xn + (size_t)(nb + k) * C;
    float *drow = d_xn + (size_t)(nb + k) * C;
    float part = 0.f;
    for (int ch = lane; ch < C; ch += 64) part += row[ch] * qr[ch];
    const float s = wave_sum(part) / denom;
    const float ds = g_out[nb + k] * gelu_grad(s) / denom;
    for (int ch = lane; ch < C; ch += 64) drow[ch] = ds * qr[ch];
    if (lane == 0) s_ds[k] = ds;
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc += s_ds[k] * xn[(size_t)(nb + k) * C + ch];
    d_q_part[(size_t)g * C + ch] = acc;     // the caller adds row g into d q[r(g)]
  }
}

static int tb_block(int C) { return C <= 64 ? 64 : (C <= 128 ? 128 : 256); }

}  // namespace isg

using namespace isg;

static int tb_check(int64_t B, int32_t C) {
  if (B < 0 || C <= 0) return ISG_EINVAL;
  if (C > TB_CCAP || B >= (1ll << 31)) return ISG_EUNSUPPORTED;
  return ISG_OK;
}

extern "C" int isg_instr_attn_graphnorm_residual_bwd(const float *ins, const float *c, const float *h, const int32_t *ptr,
                                                     const float *weight, const float *bias, const float *mean_scale,
                                                     double eps, const float *node_mask, const float *grad_out,
                                                     float *d_ins, float *d_c, float *d_h, float *d_mask,
                                                     float *partial, int64_t B, int32_t C, void *stream) {
  int st = tb_check(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!ins || !c || !h || !ptr || !weight || !bias || !mean_scale || !grad_out || !d_ins || !d_c || !d_h || !partial)
    return ISG_EINVAL;
  tail_bwd_kernel<<<(unsigned)B, tb_block(C), 0, as_stream(stream)>>>(
      ins, c, h, ptr, weight, bias, mean_scale, (float)eps, node_mask, grad_out, d_ins, d_c, d_h, d_mask, partial, C,
      (float)sqrt((double)C));
  return check_launch();
}

extern "C" int isg_global_attn_pool_bwd(const float *xn, const float *q, const int32_t *ptr, const float *node_mask,
                                        const float *grad_out, const float *grad_gate, float *d_xn, float *d_q,
                                        float *d_mask, int64_t B, int32_t C, void *stream) {
  int st = tb_check(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!xn || !q || !ptr || !grad_out || !d_xn || !d_q) return ISG_EINVAL;
  pool_bwd_kernel<<<(unsigned)B, tb_block(C), 0, as_stream(stream)>>>(xn, q, ptr, node_mask, grad_out, grad_gate, d_xn,
                                                                      d_q, d_mask, C, sqrtf((float)C));
  return check_launch();
}

extern "C" int isg_instr_gate_bwd(const float *x, const float *instr, const int32_t *ptr, const float *grad_out,
                                  float *d_x, float *d_instr, int64_t B, int32_t C, void *stream) {
  int st = tb_check(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!x || !instr || !ptr || !grad_out || !d_x || !d_instr) return ISG_EINVAL;
  instr_gate_bwd_kernel<<<(unsigned)B, tb_block(C), 0, as_stream(stream)>>>(x, instr, ptr, grad_out, d_x, d_instr, C);
  return check_launch();
}

extern "C" int isg_node_gate_bwd(const float *xn, const float *q, const int64_t *batch, int32_t double_index,
                                 const int32_t *ptr, const float *grad_gate, float *d_xn, float *d_q_partial, int64_t N,
                                 int64_t B, int32_t C, void *stream) {
  int st = tb_check(B, C);
  if (st != ISG_OK) return st;
  if (B == 0) return ISG_OK;
  if (!xn || !q || !batch || !ptr || !grad_gate || !d_xn || !d_q_partial) return ISG_EINVAL;
  node_gate_bwd_kernel<<<(unsigned)B, tb_block(C), 0, as_stream(stream)>>>(xn, q, batch, double_index, ptr, grad_gate,
                                                                           d_xn, d_q_partial, (int)N, C, sqrtf((float)C));
  return check_launch();
}
