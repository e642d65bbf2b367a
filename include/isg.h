/*
 * isg.h -- C ABI of libisg_hip.so: the MI355X (gfx950) kernels behind the ISubGVQA
 * inference hot path (batched scene-graph message passing + discrete subgraph sampling).
 *
 * The reference (DigitalPhonetics/Intrinsic-Subgraph-Generation-for-VQA) has no FFI: its
 * boundary is the Python nn.Module surface under ISubGVQA/models, and all sparse arithmetic
 * is delegated to torch_geometric / torch_scatter kernels.  Each entry point below replaces
 * one such delegated step; the comment above it cites the reference call site (file:line,
 * relative to the reference root) whose arithmetic it reproduces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named *_host
 *   - the caller owns every buffer (inputs, outputs, workspace); the library never allocates,
 *     frees or synchronises, so every call is hipGraph-capturable
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream)
 *   - return value: ISG_OK (0) or a negative ISG_E* status; nothing throws across the ABI;
 *     isg_status_string() names a status
 *   - features are fp32 row-major; PyG index tensors (edge_index, batch) arrive as int64 and
 *     are converted once by isg_graph_ptr / isg_csr_build into int32 plan arrays
 *   - N nodes, E edges, B graphs of the batch; H heads, C channels per head; 4 | C required
 */
#ifndef ISG_H
#define ISG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISG_ABI_VERSION 20

#define ISG_OK 0
#define ISG_EINVAL (-1)       /* null pointer / negative size / inconsistent sizes          */
#define ISG_EUNSUPPORTED (-2) /* shape outside what the kernels are built for (see each fn) */
#define ISG_ELAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch            */
#define ISG_EWORKSPACE (-4)   /* workspace too small                                        */

int isg_abi_version(void);
const char *isg_status_string(int status);
/* text of the last HIP error seen by a launch on this thread ("" if none) */
const char *isg_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * Graph plan (built once per PyG Batch, reused by every layer)
 * ------------------------------------------------------------------------------------------- */

/* ptr[b] = first node of graph b, ptr[B] = N, from the sorted PyG `batch` vector; nmax[0] = the
 * largest node count of any graph.  Replaces `size = batch[-1].item()+1` and the counting inside
 * to_dense_batch (ISubGVQA/models/masking.py:135,162; att_pooling.py:60).
 * batch int64[N] (sorted ascending), ptr int32[B+1], nmax int32[1]. */
int isg_graph_ptr(const int64_t *batch, int64_t N, int64_t B, int32_t *ptr, int32_t *nmax, void *stream);

/* The whole plan of a batch in six launches: ptr / bounds[0] as isg_graph_ptr, rowptr / eid / src / dst as isg_csr_build, eptr /
 * bounds[1] as isg_graph_edge_ptr (those three entry points together take fourteen launches of 4-5 us: 6 % of a BASELINE
 * configs[1] step).  bounds int32[2] = {largest node count, largest edge count of a graph}; bounds_host: NULL, or int32[2] in
 * pinned host memory the device can address (hipHostMalloc): the last kernel stores the bounds there too, so a caller that only
 * verifies size hints later needs no device->host copy in the stream (4 us + two engine switches); workspace as isg_csr_build. */
int isg_graph_plan_build(const int64_t *batch, const int64_t *edge_index, int64_t N, int64_t E, int64_t B, int32_t *ptr,
                         int32_t *bounds, int32_t *bounds_host, int32_t *rowptr, int32_t *eid, int32_t *src, int32_t *dst,
                         int32_t *eptr, void *workspace, size_t workspace_bytes, void *stream);

/* eptr[g] = rowptr[ptr[g]] for g = 0..B: the CSR-slot range of graph g when its nodes are contiguous and its edges
 * stay inside it (PyG Batch layout).  Lets the per-graph message-passing kernel read a graph's node range and edge
 * range in one round of loads.  ptr int32[B+1], rowptr int32[N+1], eptr int32[B+1]; emax int32[1] (optional)
 * receives the largest edge count of any graph. */
int isg_graph_edge_ptr(const int32_t *ptr, const int32_t *rowptr, int64_t B, int32_t *eptr, int32_t *emax,
                       void *stream);

/* Bytes of workspace isg_csr_build needs. */
size_t isg_csr_workspace_bytes(int64_t N, int64_t E);

/* CSR by destination node of the COO `edge_index` int64[2,E] (row 0 = source j, row 1 = target i,
 * flow source_to_target as in PyG MessagePassing; ISubGVQA/models/mgat_v2_conv.py:47,215).
 * Within a destination segment edges keep ascending original edge id, so every segmented
 * reduction accumulates in the order torch_scatter's CPU kernels do.
 * rowptr int32[N+1]; eid int32[E] original edge id per CSR slot; src int32[E] source node per slot;
 * dst int32[E] (optional, may be NULL) destination node per slot, used by the per-graph message-passing kernel. */
int isg_csr_build(const int64_t *edge_index, int64_t N, int64_t E, int32_t *rowptr, int32_t *eid,
                  int32_t *src, int32_t *dst, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Message passing
 * ------------------------------------------------------------------------------------------- */

/* out[n,:] = gelu(x[n,:] * instr[batch[n],:])      ISubGVQA/models/mgat_v2_conv.py:156-157
 * x,out fp32[N,C]; instr fp32[B,C]; batch int64[N].  out may alias x. */
int isg_instr_gate(const float *x, const float *instr, const int64_t *batch, float *out, int64_t N,
                   int32_t C, void *stream);

/* The same gate written as isg_gatv2_layer_conv reads its input:   ISubGVQA/models/mgat_v2_conv.py:156-157
 * planes uint16 [N][2][128] = gelu(x * instr[batch]) as per-row-scaled (hi, mid) fp16 planes (row: 128 hi values, then 128 mid
 * values; the row scale / split of isg_edge_planes), inv_scale fp32 [N]; out fp32 [N,128] or NULL (the fp32 rows as well: a
 * masked layer's node gate reads them).  ISG_EUNSUPPORTED unless C == 128. */
int isg_instr_gate_planes(const float *x, const float *instr, const int64_t *batch, float *out, uint16_t *planes,
                          float *inv_scale, int64_t N, int32_t C, void *stream);

/* out = cat((a, b, a * b), dim=1), rowmax[m] = max |out[m,:]|      ISubGVQA/models/isubgvqa.py:288-291 (the classifier head's
 * input; the maxima are the row scales' input of the exact-split Linear that reads it).  a, b fp32 [M,C]; out fp32 [M,3C];
 * rowmax fp32 [M].  ISG_EUNSUPPORTED unless C % 4 == 0. */
int isg_cat_mul_rowmax(const float *a, const float *b, float *out, float *rowmax, int64_t M, int32_t C, void *stream);

/* edge_mask[e] = mask[src[e]] * mask[dst[e]]        ISubGVQA/sampling/node_edge_masks.py:7-10
 * mask fp32[N]; edge_index int64[2,E]; out fp32[E]. */
int isg_node_to_edge_mask(const float *node_mask, const int64_t *edge_index, int64_t E, float *out,
                          void *stream);

/* GATv2 message + softmax + aggregate: MaskingGATv2Conv.message and the PyG 'add' aggregation,
 * ISubGVQA/models/mgat_v2_conv.py:243-279 (+ the bias add of :231-232 when bias != NULL).
 *   s      = x_r[i] + x_l[j] + e_proj[e];  s *= m_e;  s = leaky_relu(s, slope);  s *= m_e
 *   a[e,h] = <s[h,:], att[h,:]>
 *   alpha  = exp(a - max_i) / (sum_i exp(a - max_i) + 1e-16)      (torch_geometric.utils.softmax)
 *   out[i] = sum_e x_l[j] * alpha[e] * m_e  (+ bias)
 * m_e = edge_mask[e] if given, else node_mask[j]*node_mask[i] if given, else 1.
 * x_l,x_r fp32[N,H*C]; e_proj fp32[E,H*C] (ORIGINAL edge order); att fp32[H*C]; bias fp32[H*C]|NULL;
 * rowptr/eid/src from isg_csr_build; out fp32[N,H*C]; alpha fp32[E,H] (ORIGINAL edge order).
 * ld_l / ld_r / ld_e: row stride of x_l / x_r / e_proj in floats (0 = dense H*C); lets x_l and x_r be column slices
 * of one fused [N, 2*H*C] projection (lin_l and lin_r share their input) and e_proj a slice of one [E, L*H*C]
 * projection (every layer's lin_edge reads the same edge_attr, mgat.py:144-148).
 * H in {1,2,4,8}; 4 | C; C/4 <= 8*(64/H).  Isolated targets get 0 (+bias).
 * graph_ptr (optional, int32[B+1] from isg_graph_ptr) together with graph_eptr (isg_graph_edge_ptr), dst (int32[E]
 * from isg_csr_build) and host bounds nmax_host / emax_host (max nodes / edges of any graph, > 0) selects the
 * per-graph kernel that keeps a graph's x_l rows in LDS (x_l, x_r, out touch HBM once per row).  It requires the
 * PyG batch layout (a graph's nodes contiguous, edges inside their graph) and is used only when every graph fits its
 * LDS tables (<= 256 nodes, <= 1024 edges per graph); otherwise, or with graph_ptr == NULL, the node-chunk
 * kernel runs.  Both compute the same function (alpha to ~1e-6 relative: hardware exp2/rcp in the per-graph form). */
int isg_gatv2_mp_fwd(const float *x_l, const float *x_r, const float *e_proj, const float *att,
                     const float *bias, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                     const float *node_mask, const float *edge_mask, float *out, float *alpha, int64_t N,
                     int64_t E, int32_t H, int32_t C, float negative_slope, const int32_t *graph_ptr,
                     const int32_t *graph_eptr, const int32_t *dst, int64_t B, int32_t nmax_host,
                     int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e, void *stream);

/* Same operator with fp16 FEATURE ROWS (BASELINE configs[4], SURVEY §8d "fp16 features / fp32 accumulate"): x_l, x_r,
 * e_proj and out hold IEEE half, everything else (att, bias, masks, alpha, all arithmetic) stays fp32; out is rounded to
 * nearest even once.  ld_* count halves.  Only the per-graph kernel has this form: the batch must fit its tables
 * (ISG_EUNSUPPORTED otherwise). */
int isg_gatv2_mp_fwd_f16(const uint16_t *x_l, const uint16_t *x_r, const uint16_t *e_proj, const float *att,
                         const float *bias, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                         const float *node_mask, const float *edge_mask, uint16_t *out, float *alpha, int64_t N,
                         int64_t E, int32_t H, int32_t C, float negative_slope, const int32_t *graph_ptr,
                         const int32_t *graph_eptr, const int32_t *dst, int64_t B, int32_t nmax_host,
                         int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e, void *stream);

/* out[i,:] = sum_{e: dst(e)=i} msg[e,:] / max(deg(i),1)       torch_scatter.scatter_mean at
 * ISubGVQA/models/scene_graph_encoder.py:141.  msg fp32[E,C] (original edge order); out fp32[N,C]. */
int isg_scatter_mean(const float *msg, const int32_t *rowptr, const int32_t *eid, float *out, int64_t N,
                     int32_t C, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Node gate + discrete top-k samplers
 * ------------------------------------------------------------------------------------------- */

/* gate[n] = gelu( <xn[n,:], q[r(n),:]> / sqrt(C) )           ISubGVQA/models/masking.py:151-155
 * r(n) = batch[batch[n]] when double_index != 0 (the reference indexes an already-gathered
 * tensor again: mgat_v2_conv.py:167 + masking.py:152), else batch[n].
 * xn fp32[N,C] (= node_nn(x)); q fp32[Bq,C] (= ques_nn(u)); gate fp32[N]. */
int isg_node_gate(const float *xn, const float *q, const int64_t *batch, int32_t double_index, float *gate,
                  int64_t N, int32_t C, void *stream);

/* The same gate with node_nn inside, from the layer input as planes:   ISubGVQA/models/masking.py:137, 151-155
 *   gate[n] = gelu( < gelu(node_nn.0(x))[n,:], q[r(n),:] > / sqrt(C) )
 * x_planes uint16 [N][2][128] + x_inv_scale fp32 [N]: the gated layer input as isg_gatv2_layer_conv reads it (isg_instr_gate_planes,
 * isg_mgat_dense_tail's xp_out); w_frag / w_inv_scale = isg_split_f16x2_frag of node_nn.0.weight [128,128], b fp32 [128] its bias;
 * q, batch, double_index, gate as isg_node_gate.  The Linear runs on the fp16 three-product form (isg_linear_f16x3's arithmetic),
 * the reduction against q is isg_node_gate's; neither gelu(node_nn(x)) [N,128] nor an fp32 copy of x exists in memory.
 * ISG_EUNSUPPORTED unless C == 128. */
int isg_node_gate_planes(const uint16_t *x_planes, const float *x_inv_scale, const uint16_t *w_frag, const float *w_inv_scale,
                         const float *b, const float *q, const int64_t *batch, int32_t double_index, float *gate, int64_t N,
                         int32_t C, void *stream);

/* Row layout shared by the samplers.  Row b has `Nmax` slots:
 *   ragged input  (ptr != NULL): slot j < n_b reads scores[ptr[b]+j], slots n_b..Nmax-1 are the
 *                 0.0 pads of to_dense_batch (masking.py:162) and DO compete; outputs are written
 *                 for j < n_b at out[ptr[b]+j]               (masking.py:170-176  `[mask]`)
 *   dense input   (ptr == NULL): scores/out are [B,Nmax] row-major, all slots real.
 * Nmax = *nmax_dev if nmax_dev != NULL else nmax_host; nmax_host must be >= the true Nmax (it sizes
 * the launch); rows longer than 1024 slots are ISG_EUNSUPPORTED.
 * noise: explicit fp32 noise laid out [B, nmax_host] (the reference draws [B,Nmax] for Gumbel,
 * gumbel_scheme.py:65-69, and [B,1,Nmax,1] for I-MLE/AIMLE, wrapper.py:84-91, aimle.py:93-106 --
 * the same memory layout), or NULL to generate it in-kernel with Philox4x32-10 keyed by
 * (seed, g, j) with g = graph_ids[b] if graph_ids != NULL (int32[B]: a sub-batch whose rows keep the noise
 * streams of the graphs they were cut from, ops.run_split) else g = b. */

/* Relaxed Gumbel top-k with straight-through hard mask: GumbelSampler.forward, policy
 * 'edge_candid', ISubGVQA/sampling/methods/gumbel_scheme.py:55-58,63-104.
 *   flat = scores + g;  repeat min(k,Nmax) times { flat += log(max(1-onehot, FLT_MIN));
 *   onehot = softmax(flat/tau); khot += onehot };  hard = top-k(khot);  out = (hard-khot)+khot
 * khot_out (optional) fp32[B,Nmax_host-strided] receives khot for inspection. */
int isg_topk_gumbel(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax_host,
                    const int32_t *nmax_dev, const float *noise, uint64_t seed, const int32_t *graph_ids,
                    int32_t k, float tau, float *out, float *khot_out, void *stream);

/* Threshold top-k used by I-MLE / AIMLE at inference:  select_from_edge_candidates,
 * ISubGVQA/sampling/methods/deterministic_scheme.py:36-43, applied to scores + noise*noise_scale
 * (wrapper.py:93-100 with temperature 0 for I-MLE -> pass noise_scale = 0 and noise = NULL;
 * aimle.py:109-117 with temperature tau for AIMLE).  out = (v >= k-th largest v) as 0/1;
 * k >= Nmax -> all ones.  With noise == NULL and noise_scale != 0 the in-kernel draw is
 * Gumbel(0, 0.3) (masking.py:236,273) times noise_scale.
 * dense_out (optional) fp32[B, nmax_host]: the selection of every slot of the padded row, pads included, zero beyond
 * the batch's longest graph (the AIMLE backward counts flipped slots over the whole padded row, target_aimle.py:137). */
int isg_topk_threshold(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax_host,
                       const int32_t *nmax_dev, const float *noise, float noise_scale, uint64_t seed,
                       const int32_t *graph_ids, int32_t k, float *out, float *dense_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-graph attention / normalisation / pooling
 * ------------------------------------------------------------------------------------------- */

/* out[n,:] = softmax_{n in g}( <query[g,:], key[n,:]> / sqrt(C) ) * value[n,:]
 * ISubGVQA/utils/scatter_scaled_dot_product.py:6-15 (torch_scatter.scatter_softmax: no epsilon). */
int isg_scatter_attention(const float *query, const float *key, const float *value, const int32_t *ptr,
                          float *out, int64_t B, int32_t C, void *stream);

/* PyG GraphNorm forward: y = w*(x - mean_g*ms)/sqrt(var_g+eps) + b, per graph and channel
 * (ISubGVQA/models/mgat.py:93-95,171); eps is rounded to fp32 on the fp32 path, as `var + self.eps`
 * does.  accumulate_fp64 != 0 does all arithmetic in double and rounds
 * once at the end, as the reference does for the scene-graph encoder
 * (ISubGVQA/models/scene_graph_encoder.py:99-102). */
int isg_graph_norm(const float *x, const int32_t *ptr, const float *weight, const float *bias,
                   const float *mean_scale, double eps, int32_t accumulate_fp64, float *out, int64_t B,
                   int32_t C, void *stream);

/* One MGAT layer tail, fused:  ISubGVQA/models/mgat.py:168-177
 *   c = scatter_attention(ins, c, c);  c = GraphNorm(c);  h_out = c + h;  h_out *= node_mask (optional)
 * ins fp32[B,C]; c,h,h_out fp32[N,C] (h_out may alias h); node_mask fp32[N]|NULL. */
int isg_instr_attn_graphnorm_residual(const float *ins, const float *c, const float *h, const int32_t *ptr,
                                      const float *weight, const float *bias, const float *mean_scale,
                                      double eps, const float *node_mask, float *h_out, int64_t B, int32_t C,
                                      void *stream);

/* Tile plan of the fused per-layer kernels: consecutive graphs packed greedily into tiles of at most node_cap nodes (and,
 * when eptr != NULL, edge_cap CSR slots); a graph larger than a cap gets a tile of its own (callers test the batch's bounds).
 * tile_ptr int32[capacity + 1] receives the first graph of every tile and B behind the last; ntiles int32[1] the count (it
 * stays on the device: kernels launch `capacity` workgroups, or walk the tiles persistently); tile_info (optional, 16-byte
 * aligned int32[capacity * 4]) receives {first node, nodes, first CSR slot, CSR slots} per tile, so that a workgroup learns its
 * tile in one load.  isg_tile_plan_capacity() bounds the count from the sizes alone.  No reference counterpart: the reference's kernels are per-op, not per-layer.  * tile_info_heavy_first (NULL, or int32 [capacity, 4], 16-byte aligned, tile_info requested too): the same descriptors ordered
 * by descending CSR-slot count (in 32-slot classes, ties in tile order).  isg_gatv2_layer_conv / isg_gatv2_tile_conv walk their
 * tile list persistently, workgroup w taking entries w, w + G, ...: handed THIS list they get one tile of every weight class per
 * round (the slowest workgroup's share 1.03x the mean instead of 1.06x at BASELINE configs[1]); any order gives the same results. */
int64_t isg_tile_plan_capacity(int64_t N, int64_t E, int64_t B, int32_t node_cap, int32_t edge_cap);
int isg_tile_plan(const int32_t *ptr, const int32_t *eptr, int64_t B, int32_t node_cap, int32_t edge_cap,
                  int32_t *tile_ptr, int32_t *ntiles, int32_t *tile_info, int64_t capacity, int32_t *tile_info_heavy_first,
                  void *stream);

/* The dense back half of one MGAT layer and the first line of the next as ONE launch on graph-aligned 64-row tiles
 * (isg_tile_plan with node_cap = 64):   ISubGVQA/models/mgat.py:156-177, mgat_v2_conv.py:156-157
 *   c = gelu(x_proj.2(gelu(x_proj.0(conv_out))));  c = scatter_attention(ins, c, c);  c = GraphNorm(c);
 *   h_out = c + h;  h_out *= node_mask (optional);  xg = gelu(h_out * ins_next[batch]) (optional: xg_out fp32 [N,C] and / or
 *   xp_out uint16 [N][2][128] + xinv_out fp32 [N] = the same rows as the scaled (hi, mid) planes isg_gatv2_layer_conv reads)
 * conv_out fp32[N,K1] (row stride lda) with a_rowmax fp32[N,P] (row stride ldp) = partial maxima of |conv_out| per row (what
 * isg_gatv2_mp_fwd_rowmax / _logits leave); w1 / w2: isg_split_f16x2_frag planes + inverse scales of x_proj.0.weight [MID,K1]
 * and x_proj.2.weight [C,MID]; b1 fp32[MID], b2 fp32[C]; y_bound fp32[2] = {max_j sum_k |w1[j,k]|, max_j |b1[j]|} (the
 * intermediate's row scales come from |x_proj.0(a)_ij| <= max_k |a_ik| * y_bound[0] + y_bound[1]); ins / ins_next fp32[B,C]; h, h_out, xg_out fp32[N,C] (h_out must not
 * alias h); batch int64[N].  Arithmetic of the tail: isg_instr_attn_graphnorm_residual's.  ISG_EUNSUPPORTED unless
 * K1 = 512, MID = 256, C = 128 (BASELINE configs[1]); graphs beyond 64 nodes are truncated to the tile (callers test
 * the batch's bound first). */
int isg_mgat_dense_tail(const float *conv_out, int32_t lda, const float *a_rowmax, int32_t P, int32_t ldp,
                        const uint16_t *w1_frag, const float *w1_inv_scale, const float *b1, const float *y_bound,
                        const uint16_t *w2_frag,
                        const float *w2_inv_scale, const float *b2, const float *ins, const float *h,
                        const float *gn_weight, const float *gn_bias, const float *gn_mean_scale, double eps,
                        const float *node_mask, const float *ins_next, float *h_out, float *xg_out, uint16_t *xp_out,
                        float *xinv_out, const int32_t *ptr, const int64_t *batch, const int32_t *tile_ptr,
                        const int32_t *tile_info, const int32_t *ntiles, int64_t max_tiles, int64_t N, int32_t K1, int32_t MID,
                        int32_t C, void *stream);

/* MaskingGATv2Conv.message + aggregate with lin_edge inside as ONE launch on graph-aligned tiles (isg_tile_plan with node_cap =
 * 64, edge_cap = 256 and tile_info):   ISubGVQA/models/mgat_v2_conv.py:243-279 (lin_edge :259-261)
 * Persistent workgroups (two per CU) each keep one head -- its lin_edge fragments stay in registers -- and walk tiles: the head's
 * x_l slice of a tile's nodes is staged once in LDS and serves the logit epilogue of the edge GEMM (row gathers) and the
 * aggregation; logits never leave LDS; the next tile's inputs are requested while the current one is finished.  Operands as
 * isg_gatv2_edge_logits / isg_gatv2_mp_fwd_logits (x_l / x_r fp32 [N,H*C] with row strides ldl / ldr, w_frag / w_inv_scale =
 * isg_split_f16x2_frag of lin_edge.weight [H*C,K], CSR by destination) except that the edge features arrive as edge_planes /
 * edge_inv_scale (isg_edge_planes: split once per batch, in CSR slot order, for all layers and heads); out fp32 [N,H*C] (row
 * stride ldo), alpha fp32 [E,H] in edge-id order, rowmax fp32 [N,H] or NULL.  Results are bit-identical to that pair.
 * ISG_EUNSUPPORTED unless C == 128, K <= 128, K % 4 == 0; graphs beyond a tile's caps are truncated (callers test the bounds). */
int isg_gatv2_tile_conv(const float *x_l, int32_t ldl, const float *x_r, int32_t ldr, const uint16_t *edge_planes,
                        const float *edge_inv_scale, const uint16_t *w_frag, const float *w_inv_scale, const float *att,
                        const float *bias,
                        const int32_t *rowptr, const int32_t *eid, const int32_t *src, const int32_t *dst,
                        const int32_t *tile_info, const int32_t *ntiles, int64_t max_tiles, const float *node_mask,
                        const float *edge_mask, float *out, int32_t ldo, float *alpha, float *rowmax, int64_t N, int64_t E,
                        int32_t H, int32_t C, int32_t K, float negative_slope, void *stream);

/* One MaskingGATv2Conv after its instruction gate and node mask as ONE persistent launch: lin_l | lin_r, lin_edge, logits,
 * softmax, aggregation (ISubGVQA/models/mgat_v2_conv.py:177-181, :215-232, :243-279).  As isg_gatv2_tile_conv, but the head's x_l
 * and x_r slices of a tile are formed inside, on the matrix cores, from the gated node rows -- which arrive as scaled (hi, mid)
 * planes x_planes uint16 [N][2][128] + x_inv_scale fp32 [N] (isg_instr_gate_planes, isg_mgat_dense_tail's xp_out, or
 * isg_edge_planes with eid = NULL on fp32 rows: the split is made once per row, not once per tile and head) -- and
 * stay in LDS: neither tensor exists in memory, the logit epilogue gathers both from LDS.  wn_frag / wn_inv_scale =
 * isg_split_f16x2_frag of cat(lin_l.weight, lin_r.weight) [2*H*C,128]; bn fp32 [2*H*C] = cat of their biases; the remaining
 * operands as isg_gatv2_tile_conv.  Bit-identical to isg_linear_f16x3 + isg_gatv2_tile_conv.  ISG_EUNSUPPORTED unless C == 128,
 * K_in == 128, K_edge <= 128, K_edge % 4 == 0, H <= 16. */
int isg_gatv2_layer_conv(const uint16_t *x_planes, const float *x_inv_scale, const uint16_t *wn_frag, const float *wn_inv_scale,
                         const float *bn,
                         const uint16_t *edge_planes, const float *edge_inv_scale, const uint16_t *we_frag,
                         const float *we_inv_scale, const float *att, const float *bias, const int32_t *rowptr,
                         const int32_t *eid, const int32_t *src, const int32_t *dst, const int32_t *tile_info,
                         const int32_t *ntiles, int64_t max_tiles, const float *node_mask, const float *edge_mask, float *out,
                         int32_t ldo, float *alpha, float *rowmax, int64_t N, int64_t E, int32_t H, int32_t C, int32_t K_in,
                         int32_t K_edge, float negative_slope, void *stream);

/* GlobalAttention.forward on graph-aligned 64-row tiles as one launch:   ISubGVQA/models/att_pooling.py:57-77
 *   xn = node_nn(x) (Linear GELU Linear) * node_mask;  gate = softmax_g(<xn_n, q_g> / sqrt(C)) (+1e-16);  out_g = sum_n gate_n xn_n
 * x fp32 [N,128] (row stride ldx); w1 / w2 = isg_split_f16x2_frag planes + inverse scales of node_nn.0 / node_nn.2 weights
 * [128,128], b1 / b2 their biases; y_bound fp32 [2] = {max_j sum_k |w1[j,k]|, max_j |b1[j]|} (row scales of the intermediate, as
 * in isg_mgat_dense_tail); q fp32 [B,128] = ques_nn(u); node_mask fp32 [N] or NULL; out fp32 [B,128], gate fp32 [N]; tiles from
 * isg_tile_plan (node_cap = 64, tile_info requested).  The per-graph arithmetic is isg_global_attn_pool's.  ISG_EUNSUPPORTED
 * unless C == 128. */
int isg_readout_tile(const float *x, int32_t ldx, const uint16_t *w1_frag, const float *w1_inv_scale, const float *b1,
                     const float *y_bound, const uint16_t *w2_frag, const float *w2_inv_scale, const float *b2, const float *q,
                     const float *node_mask, float *out, float *gate, const int32_t *ptr, const int64_t *batch,
                     const int32_t *tile_ptr, const int32_t *tile_info, const int32_t *ntiles, int64_t max_tiles, int64_t N,
                     int32_t C, void *stream);

/* Edge features as the exact-split kernels want them, once per batch: per-row power-of-two scale and (hi, mid) fp16 planes in CSR
 * SLOT order (slot t = edge eid[t]): planes uint16 [E][2][128] (row: 128 hi values, then 128 mid values; columns beyond K zero),
 * inv_scale fp32 [E].  edge_attr fp32 [E,K] by edge id (row stride lda).  The same edge features feed lin_edge of every layer
 * (ISubGVQA/models/mgat.py:144-148).  eid == NULL: rows in their own order (any fp32 [E,K] matrix, e.g. node rows for
 * isg_gatv2_layer_conv).  ISG_EUNSUPPORTED unless K <= 128, K % 4 == 0. */
int isg_edge_planes(const float *edge_attr, int32_t lda, const int32_t *eid, int64_t E, int32_t K, uint16_t *planes,
                    float *inv_scale, void *stream);

/* isg_tile_plan and isg_edge_planes as ONE launch: both depend on the graph plan only, and the tile plan is a single workgroup
 * (22 us at 4096 graphs) that otherwise has the chip to itself; here it is workgroup 0 of the launch that splits the edge rows.
 * Operands and results exactly as the two entry points'. */
int isg_tile_plan_edge_planes(const int32_t *ptr, const int32_t *eptr, int64_t B, int32_t node_cap, int32_t edge_cap,
                              int32_t *tile_ptr, int32_t *ntiles, int32_t *tile_info, int64_t capacity,
                              int32_t *tile_info_heavy_first, const float *edge_attr, int32_t lda, const int32_t *eid, int64_t E,
                              int32_t K, uint16_t *planes, float *inv_scale, void *stream);

/* Question-conditioned softmax pooling: GlobalAttention.forward, ISubGVQA/models/att_pooling.py:63-73
 *   x = xn * node_mask;  gate = softmax_g(<x, q[g]>/sqrt(C)) (+1e-16 in the denominator);
 *   out[g,:] = sum_n gate[n] * x[n,:]
 * xn fp32[N,C] (= node_nn(x)); q fp32[B,C] (= ques_nn(u)); node_mask fp32[N]|NULL;
 * out fp32[B,C]; gate fp32[N]. */
int isg_global_attn_pool(const float *xn, const float *q, const int32_t *ptr, const float *node_mask,
                         float *out, float *gate, int64_t B, int32_t C, void *stream);

/* SIMPLE sampler: EdgeSIMPLEBatched.forward, policy 'edge_candid', one sample,
 * ISubGVQA/sampling/methods/simple_scheme.py:44-162 over Layer.log_pr / sample (simple.py:203-251) and the exactly-k
 * circuit of create_simple_constraint.py:34-73.  Per row: scores padded with 0.0 up to nmax (to_dense_batch) and with
 * -1e10 up to n = 2^ceil(log2 nmax); marginals = exp(log-marginals of the circuit's positive literals), including the
 * reference's dummy-node padding of element and parent lists; sample = k largest of scores - log(-log(u));
 * out = (sample - marginals) + marginals.  nmax must be the batch's TRUE longest row (it fixes the circuit).
 * scores ragged [N] with ptr or dense [B,nmax]; uniform fp32[B,n] (the torch.rand draw) or NULL for the in-kernel
 * Philox stream of `seed`; out in the layout of scores; marg_out optional fp32[B,nmax].  k <= 16, nmax <= 1024. */
int isg_simple_topk(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax, const float *uniform,
                    uint64_t seed, const int32_t *graph_ids, int32_t k, float *out, float *marg_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Backward (training of the hot path)
 * ------------------------------------------------------------------------------------------- */

/* Backward of isg_gatv2_mp_fwd with respect to x_l, x_r, e_proj, att and (optionally) the edge mask.  Dense rows only
 * (ld = H*C).  alpha is the forward's attention output; grad_out fp32[N,H*C] is d out (the bias gradient is its column
 * sum, left to the caller).  rowptr/eid/src: CSR by destination; rowptr_s/eid_s/dst_s: CSR by source, i.e.
 * isg_csr_build on the flipped edge_index (its `src` output is then the destination per slot).
 * d_att_partial fp32[ceil(N/16), H*C]: per-workgroup partial sums, d att = their column sum (fixed order, no atomics).
 * d_edge_mask fp32[E] or NULL.  The gradient flowing into alpha is not supported. */
int isg_gatv2_mp_bwd(const float *x_l, const float *x_r, const float *e_proj, const float *att, const float *alpha,
                     const float *grad_out, const int32_t *rowptr, const int32_t *eid, const int32_t *src,
                     const int32_t *rowptr_s, const int32_t *eid_s, const int32_t *dst_s, const float *node_mask,
                     const float *edge_mask, float *d_x_l, float *d_x_r, float *d_e_proj, float *d_att_partial,
                     float *d_edge_mask, int64_t N, int64_t E, int32_t H, int32_t C, float negative_slope, void *stream);

/* Straight-through backward of isg_topk_gumbel (gumbel_scheme.py:83-90): d_out is the gradient of the mask in the
 * forward's layout (ragged [N] with ptr, dense [B,nmax] without), d_scores the gradient of the scores, same layout.
 * scores / noise / seed / k / tau must be the forward's.  ISG_EUNSUPPORTED when k * row length exceeds the 64 KB LDS
 * history (k * nmax_slots > 4096). */
int isg_topk_gumbel_bwd(const float *scores, const int32_t *ptr, int64_t B, int32_t nmax_host,
                        const int32_t *nmax_dev, const float *noise, uint64_t seed, const int32_t *graph_ids,
                        int32_t k, float tau, const float *d_out, float *d_scores, void *stream);

/* Backward of isg_instr_attn_graphnorm_residual (autograd over mgat.py:168-177 in the reference).  grad_out fp32[N,C].
 * d_ins fp32[B,C], d_c / d_h fp32[N,C], d_mask fp32[N] or NULL; partial fp32[B,3,C]: per-graph rows of
 * (d weight, d bias, d mean_scale), summed over B by the caller (fixed order, no atomics). */
int isg_instr_attn_graphnorm_residual_bwd(const float *ins, const float *c, const float *h, const int32_t *ptr,
                                          const float *weight, const float *bias, const float *mean_scale, double eps,
                                          const float *node_mask, const float *grad_out, float *d_ins, float *d_c,
                                          float *d_h, float *d_mask, float *partial, int64_t B, int32_t C, void *stream);

/* Backward of isg_global_attn_pool (att_pooling.py:63-73).  grad_out fp32[B,C]; grad_gate fp32[N] or NULL;
 * d_xn fp32[N,C], d_q fp32[B,C], d_mask fp32[N] or NULL. */
int isg_global_attn_pool_bwd(const float *xn, const float *q, const int32_t *ptr, const float *node_mask,
                             const float *grad_out, const float *grad_gate, float *d_xn, float *d_q, float *d_mask,
                             int64_t B, int32_t C, void *stream);

/* Backward of isg_instr_gate (mgat_v2_conv.py:156-157) for a sorted batch: ptr int32[B+1] from isg_graph_ptr.
 * d_x fp32[N,C], d_instr fp32[B,C]. */
int isg_instr_gate_bwd(const float *x, const float *instr, const int32_t *ptr, const float *grad_out, float *d_x,
                       float *d_instr, int64_t B, int32_t C, void *stream);

/* Backward of isg_node_gate (masking.py:151-155).  grad_gate fp32[N]; d_xn fp32[N,C]; d_q_partial fp32[B,C]: row g is
 * the contribution of graph g's nodes to d q[r(g)], r(g) = batch[g] with double_index (quirk Q3) else g; the caller
 * scatters the rows (several graphs may share one row of q). */
int isg_node_gate_bwd(const float *xn, const float *q, const int64_t *batch, int32_t double_index, const int32_t *ptr,
                      const float *grad_gate, float *d_xn, float *d_q_partial, int64_t N, int64_t B, int32_t C,
                      void *stream);

/* Weight gradient of a linear layer, dW[N,K] = grad_out^T x (grad_out fp32[M,N] row stride ldg, x fp32[M,K] row stride
 * ldx): a split-M GEMM on the fp32 matrix-core instruction.  partial fp32[splits,N,K] receives one partial result per
 * split (every element is written); dW is their sum over the first axis, left to the caller (fixed order).
 * isg_linear_wgrad_splits returns the recommended number of splits for a shape. */
int64_t isg_linear_wgrad_splits(int64_t M, int32_t N, int32_t K);
int isg_linear_wgrad(const float *grad_out, const float *x, float *partial, int64_t M, int32_t N, int32_t K,
                     int32_t ldg, int32_t ldx, int64_t splits, void *stream);

/* NodeMaskToEdgeMask.backward, ISubGVQA/sampling/node_edge_masks.py:13-19: d_node_mask[i] = sum over edges INTO i of
 * d_edge_mask[e] (the reference's rule: destination only, no product rule).  CSR by destination. */
int isg_node_to_edge_mask_bwd(const float *d_edge_mask, const int32_t *rowptr, const int32_t *eid, float *d_node_mask,
                              int64_t N, void *stream);

/* isg_gatv2_mp_fwd that also writes rowmax fp32 [N, H] = the largest |out| of every (node, head): the row scales of the
 * fp16 three-product GEMM that consumes `out` (isg_linear_f16x3_tile takes them as its a_rowmax with P = H), at no extra
 * pass over `out`.  Grouped per-graph kernel only: ISG_EUNSUPPORTED for batches / widths that take another kernel. */
int isg_gatv2_mp_fwd_rowmax(const float *x_l, const float *x_r, const float *e_proj, const float *att, const float *bias,
                            const int32_t *rowptr, const int32_t *eid, const int32_t *src, const float *node_mask,
                            const float *edge_mask, float *out, float *alpha, float *rowmax, int64_t N, int64_t E,
                            int32_t H, int32_t C, float negative_slope, const int32_t *graph_ptr,
                            const int32_t *graph_eptr, const int32_t *dst, int64_t B, int32_t nmax_host,
                            int32_t emax_host, int32_t ld_l, int32_t ld_r, int32_t ld_e, void *stream);

/* GATv2 attention logits without e_proj in memory (csrc/isg_mp_logits.hip): for every CSR slot t (edge eid[t] from src[t]
 * into dst[t]) and head h,  logits[t, h] = att_h . leaky_relu(x_l[src] + x_r[dst] + lin_edge(edge_attr[eid]))_h  with the
 * edge mask applied before and after the leaky_relu -- MaskingGATv2Conv.message up to the softmax
 * (mgat_v2_conv.py:243-270, lin_edge of :259-261 inside).  edge_attr fp32 rows by EDGE ID (stride lda), w_frag /
 * w_inv_scale from isg_split_f16x2_frag(lin_edge.weight [H*C, K]), x_l / x_r fp32 rows by node id (strides ldl / ldr),
 * eid / src / dst int32 in CSR slot order (isg_csr_build), edge_mask fp32 [E] by edge id or node_mask fp32 [N] (product of
 * the endpoints, NodeMaskToEdgeMask) or both NULL.  logits fp32 [E, H] in SLOT order.  The product is the fp16
 * three-term form of isg_linear_f16x3 and never leaves the accumulators.  head_stride_l / _r: distance in floats between
 * the head slices of one node (0 = C: heads side by side in a row of H*C; N*C with ldl = C for a head-major [H][N][C]
 * tensor, whose 512-byte rows keep a gather instruction inside a few pages).  A head dimension that is not a multiple of 32
 * (the reference's C = 300) runs on heads PADDED to Cp = 32 * ceil(C / 32) channels: w_frag / w_inv_scale are then those of
 * the [H*Cp, K] matrix with zero rows behind every head's C-th; att, x_l, x_r stay unpadded.  K = 128 and 128 < K <= 304 run the rows
 * kernel (a wave keeps its 32 slots' edge rows in registers, the weight tiles stream through LDS once per 224 slots, requested by a wave of their own): for
 * K > 128 w_frag / w_inv_scale are those of the [H*Cp, 304] matrix (zero columns behind the K-th).  ISG_EUNSUPPORTED unless 4 | C, K <= 304,
 * 4 | K, H * Cp <= 2048, 16-byte aligned rows. */
int isg_gatv2_edge_logits(const float *edge_attr, int32_t lda, const uint16_t *w_frag, const float *w_inv_scale,
                          const float *x_l, int32_t ldl, int64_t head_stride_l, const float *x_r, int32_t ldr,
                          int64_t head_stride_r, const float *att, const int32_t *eid, const int32_t *src, const int32_t *dst,
                          const float *edge_mask, const float *node_mask, float *logits, int64_t E, int32_t H, int32_t C,
                          int32_t K, float negative_slope, void *stream);
/* The same with x_l / x_r as HALF rows (BASELINE configs[4]: fp16 feature rows, fp32 arithmetic; ldl / ldr / head strides count
 * halves): the edge projection is rounded to half before it enters the logit, which is how isg_linear_f16x3_f16 stores e_proj for
 * isg_gatv2_mp_fwd_f16 -- the pair then computes what the un-fused fp16 path computes.  The rows kernel only: K >= 128
 * (ISG_EUNSUPPORTED below).  Replaces mgat_v2_conv.py:259-261 + the logit half of :243-270 on half rows. */
int isg_gatv2_edge_logits_f16(const float *edge_attr, int32_t lda, const uint16_t *w_frag, const float *w_inv_scale,
                              const uint16_t *x_l, int32_t ldl, int64_t head_stride_l, const uint16_t *x_r, int32_t ldr,
                              int64_t head_stride_r, const float *att, const int32_t *eid, const int32_t *src,
                              const int32_t *dst, const float *edge_mask, const float *node_mask, float *logits, int64_t E,
                              int32_t H, int32_t C, int32_t K, float negative_slope, void *stream);
/* isg_gatv2_mp_fwd whose result leaves as the SEGMENTED planes32 operand of isg_linear_h3p instead of fp32 rows: H = 4 and a head
 * dimension served by the flat per-graph kernel (the reference's C = 300); ISG_EUNSUPPORTED wherever that kernel does not run --
 * the caller then takes isg_gatv2_mp_fwd and isg_split_planes32.  out_planes uint16 [N][2 * ceil(2 C / 32)][64]; out_inv fp32
 * [2][N]: the row scales of columns [0, 2C) (a_inv_first of isg_linear_h3p, k_split = 32 * ceil(2 C / 32)) and [2C, 4C) (a_inv).
 * x_proj.0 (ISubGVQA/models/mgat.py:156) reads it with no pass in between. */
int isg_gatv2_mp_fwd_planes(const float *x_l, const float *x_r, const float *e_proj, const float *att, const float *bias,
                            const int32_t *rowptr, const int32_t *eid, const int32_t *src, const float *node_mask,
                            const float *edge_mask, uint16_t *out_planes, float *out_inv, float *alpha, int64_t N, int64_t E,
                            int32_t H, int32_t C, float negative_slope, const int32_t *graph_ptr, const int32_t *graph_eptr,
                            const int32_t *dst, int64_t B, int32_t nmax_host, int32_t emax_host, int32_t ld_l, int32_t ld_r,
                            int32_t ld_e, void *stream);

/* The rest of MaskingGATv2Conv.message + aggregate (mgat_v2_conv.py:270-279, :215-232) given those logits: softmax over
 * every destination's in-edges (+1e-16), alpha fp32 [E, H] by edge id, out[i] = sum alpha * mask * x_l[src] + bias, rowmax
 * (optional) as in isg_gatv2_mp_fwd_rowmax.  e_proj and x_r are not read.  Per-graph kernel only: ISG_EUNSUPPORTED for
 * batches / widths it has no instantiation for (the caller then runs isg_linear_* + isg_gatv2_mp_fwd). */
int isg_gatv2_mp_fwd_logits(const float *x_l, const float *logits, const float *att, const float *bias,
                            const int32_t *rowptr, const int32_t *eid, const int32_t *src, const float *node_mask,
                            const float *edge_mask, float *out, float *alpha, float *rowmax, int64_t N, int64_t E,
                            int32_t H, int32_t C, float negative_slope, const int32_t *graph_ptr,
                            const int32_t *graph_eptr, const int32_t *dst, int64_t B, int32_t nmax_host,
                            int32_t emax_host, int32_t ld_l, void *stream);
/* The same on HALF feature rows (x_l in, out; ld_l counts halves; no row maxima): behind isg_gatv2_edge_logits_f16, BASELINE
 * configs[4]'s storage.  The grouped per-graph kernel only (mgat_v2_conv.py:270-279, :215-232). */
int isg_gatv2_mp_fwd_logits_f16(const uint16_t *x_l, const float *logits, const float *att, const float *bias,
                                const int32_t *rowptr, const int32_t *eid, const int32_t *src, const float *node_mask,
                                const float *edge_mask, uint16_t *out, float *alpha, int64_t N, int64_t E, int32_t H, int32_t C,
                                float negative_slope, const int32_t *graph_ptr, const int32_t *graph_eptr, const int32_t *dst,
                                int64_t B, int32_t nmax_host, int32_t emax_host, int32_t ld_l, void *stream);
/* The same with the result as the segmented planes32 operand of isg_linear_h3p (isg_gatv2_mp_fwd_planes's output contract:
 * H = 4, the flat per-graph kernel): a C = 300 layer with neither e_proj nor fp32 convolution rows in memory. */
int isg_gatv2_mp_fwd_logits_planes(const float *x_l, const float *logits, const float *att, const float *bias,
                                   const int32_t *rowptr, const int32_t *eid, const int32_t *src, const float *node_mask,
                                   const float *edge_mask, uint16_t *out_planes, float *out_inv, float *alpha, int64_t N,
                                   int64_t E, int32_t H, int32_t C, float negative_slope, const int32_t *graph_ptr,
                                   const int32_t *graph_eptr, const int32_t *dst, int64_t B, int32_t nmax_host,
                                   int32_t emax_host, int32_t ld_l, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Dense projections (fp32 accuracy on the bf16 matrix cores)
 * ------------------------------------------------------------------------------------------- */

/* planes[q][row][Kp] (q = 0..2, Kp = K rounded up to 32, zero padded, bf16 bit patterns) with
 * w[row][k] = planes[0] + planes[1] + planes[2] exactly: each plane is the bf16 rounding of what the previous ones
 * left.  Weights are static, so this runs once per weight.  w fp32[rows,K]; planes uint16[3*rows*Kp]. */
int isg_split_bf16x3(const float *w, int64_t rows, int32_t K, uint16_t *planes, void *stream);

/* d[M,N] = act(a[M,K] @ W[N,K]^T + bias):  torch.nn.Linear / PyG Linear (+ the GELU that follows it) as used by
 * ISubGVQA/models/mgat_v2_conv.py:177,181,259, mgat.py:156, masking.py:137,152, att_pooling.py:62,66.
 * a fp32 with row stride lda; w_planes from isg_split_bf16x3; bias fp32[N] or NULL; d fp32 with row stride ldd;
 * act 0 = none, 1 = exact (erf) GELU, 2 = ReLU (nn.TransformerEncoderLayer's FFN, question_encoder.py:22-25; fp32 rows
 * only).  Products are formed from the three bf16 planes of both operands (six MFMA
 * terms, fp32 accumulate): fp32-level accuracy (rel. rms ~1e-7) at bf16 matrix-core speed.
 * Requires 4 | K, 4 | lda, a 16-byte aligned. */
int isg_linear_bf16x6(const float *a, const uint16_t *w_planes, const float *bias, float *d, int64_t M, int32_t N,
                      int32_t K, int32_t lda, int32_t ldd, int32_t act, void *stream);

/* The same Linear for SMALL M (a handful of questions per forward: run_token_coo.py:49-79 evaluates one), where the tile kernels
 * are bound by the latency of their serial k loop (20-36 us per launch whatever the size): the reduction is split over the eight
 * waves of a workgroup, operands go from memory straight into TRUE fp32 MFMAs (v_mfma_f32_32x32x2_f32; no planes, no row scales,
 * no weight preparation), the eight partial tiles are added in wave order -- a row's bits depend on K alone, not on the batch.
 * a fp32 [M,K] (row stride lda); w fp32 [N,K] in torch's Linear layout (row stride ldw); bias fp32 [N] or NULL; d fp32 (row
 * stride ldd); act 0 none, 1 exact GELU, 2 ReLU.  ISG_EUNSUPPORTED unless 4 | K, 4 | lda, 4 | ldw, a and w 16-byte aligned.
 * csrc/isg_gemm_skinny.hip. */
int isg_linear_skinny(const float *a, int32_t lda, const float *w, int32_t ldw, const float *bias, float *d, int32_t ldd, int64_t M,
                      int32_t N, int32_t K, int32_t act, void *stream);

/* isg_linear_bf16x6 reading A as fp16 (a_is_f16) and / or writing D as fp16 (d_is_f16, one rounding after bias and
 * activation); lda / ldd count elements of the respective type.  The products are still exact: an fp16 value splits
 * into two bf16 terms. */
int isg_linear_bf16x6_f16(const void *a, int32_t a_is_f16, const uint16_t *w_planes, const float *bias, void *d,
                          int32_t d_is_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                          void *stream);

/* Row-panel form of the same Linear (csrc/isg_gemm_panel.hip; same call sites as isg_linear_bf16x6, same arithmetic:
 * three bf16 planes per operand, six MFMA terms, fp32 accumulate).  The weight planes are stored FRAGMENT-MAJOR,
 * planes[q][n/32][k/16][lane][8] (zero padded to 32 rows / 16 k), so that one matrix-core B operand is one contiguous
 * 1 KB load; isg_split_bf16x3_frag_elems gives the uint16 count to allocate.  A 64-row panel of `a` is split into its
 * planes once and serves every output column when K <= 128.  The accumulation order of an output element depends only
 * on K: a row's result does not depend on where the row sits in the batch.  a / d may be fp16 as in
 * isg_linear_bf16x6_f16.  Requires 4 | K, 4 | lda, a aligned to 16 bytes (8 for fp16). */
int64_t isg_split_bf16x3_frag_elems(int64_t rows, int32_t K);
int isg_split_bf16x3_frag(const float *w, int64_t rows, int32_t K, uint16_t *planes, void *stream);
int isg_linear_panel(const void *a, int32_t a_is_f16, const uint16_t *w_frag, const float *bias, void *d,
                     int32_t d_is_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                     void *stream);
/* Several Linears over the SAME rows as one launch: the weights are concatenated along the output dimension (N columns
 * in all, w_frag / bias of the concatenation) and output columns [o * out_cols, (o + 1) * out_cols) go to the o-th result
 * tensor at d + o * out_stride (elements), row stride ldd.  MGAT hands every layer the same edge features
 * (ISubGVQA/models/mgat.py:144-148), so the layers' lin_edge projections (mgat_v2_conv.py:259) share one pass over
 * edge_attr: a row panel is loaded and split into its bf16 planes once for all of them, and each layer still gets its
 * own dense [E, H*C] tensor.  out_cols a multiple of 32 that divides N. */
int isg_linear_panel_multi(const void *a, int32_t a_is_f16, const uint16_t *w_frag, const float *bias, void *d,
                           int32_t d_is_f16, int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act,
                           int32_t out_cols, int64_t out_stride, void *stream);

/* The same Linear for K <= 128 on the fp16 matrix cores with THREE products per term (csrc/isg_gemm_f16x3.hip): every A
 * row and every W row is scaled by its own power of two into fp16's normal range, split into two fp16 planes
 * (hi + mid = the scaled value to 2^-24), and hi_a hi_b + hi_a mid_b + mid_a hi_b is accumulated in fp32; the epilogue
 * scales back.  Accuracy of a plain fp32 GEMM (measured 1.0-1.1x its error) at half the matrix-core work of the bf16
 * six-product form.  isg_split_f16x2_frag: planes uint16[isg_split_f16x2_frag_elems(rows, K)] (fragment-major, 2 planes) and
 * inv_scale fp32[32 * ceil(rows / 32)].  out_cols / out_stride as in isg_linear_panel_multi (out_cols == N, out_stride == 0
 * for one output).  fp32 rows only; ISG_EUNSUPPORTED for K > 128. */
/* Tile form of the three-product kernel for 128 < K <= 1024 (x_proj of MGAT, mgat.py:156).  A row's scale needs the
 * row's largest magnitude over all of K, so it is an input: a_rowmax fp32 [M, P], P partial maxima per row written by the
 * kernel that produced `a` (isg_gatv2_mp_fwd_rowmax: P = H; this kernel: d_rowmax [M, ceil(N / 32)], or NULL).  w_planes /
 * w_inv_scale from isg_split_f16x2_rows: uint16[2 * rows * Kp] (Kp = K rounded up to 32), fp32[rows].  act 0 none, 1 exact
 * GELU, 2 ReLU.  Reductions longer than 1024 run as K-chunks: a call covers columns [k_offset, k_offset + K) of a weight
 * with K_total columns (`a` points at the chunk's first column, a_rowmax holds the chunk's row maxima) and, with
 * accumulate != 0, adds to what `d` holds before bias and activation -- every chunk is its own fp32 accumulation chain.
 * a_rowmax rows are ldp floats apart (ldp >= P): a K-chunk of a reduction whose producer left one partial maximum per
 * 32 (or 64, ...) columns takes the slice of partials that covers its columns, no pass of its own over `a`. */
int isg_split_f16x2_rows(const float *w, int64_t rows, int32_t K, uint16_t *planes, float *inv_scale, void *stream);
int isg_linear_f16x3_tile(const float *a, const float *a_rowmax, int32_t P, int32_t ldp, const uint16_t *w_planes,
                          const float *w_inv_scale, const float *bias, float *d, float *d_rowmax, int64_t M, int32_t N,
                          int32_t K, int32_t lda, int32_t ldd, int32_t act, int32_t K_total, int32_t k_offset,
                          int32_t accumulate, void *stream);
/* rowmax[m] = max_k |a[m, k]| (fp32 [M]): a_rowmax (P = 1) for an input whose producer left none; one pass over `a`. */
int isg_row_absmax(const float *a, int64_t M, int32_t K, int32_t lda, float *rowmax, void *stream);
int64_t isg_split_f16x2_frag_elems(int64_t rows, int32_t K);
int isg_split_f16x2_frag(const float *w, int64_t rows, int32_t K, uint16_t *planes, float *inv_scale, void *stream);
int isg_linear_f16x3(const float *a, const uint16_t *w_frag, const float *w_inv_scale, const float *bias, float *d,
                     int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act, int32_t out_cols,
                     int64_t out_stride, void *stream);
/* isg_linear_f16x3 with the result rounded once (RNE) to half rows d uint16 (ldd, out_stride in halves): a model in "fp16
 * features / fp32 accumulate" mode (BASELINE configs[4]; MaskingGATv2Conv.feature_dtype) projects x_l | x_r and e_proj with it
 * (ISubGVQA/models/mgat_v2_conv.py:177-181, :259-261). */
int isg_linear_f16x3_f16(const float *a, const uint16_t *w_frag, const float *w_inv_scale, const float *bias, uint16_t *d,
                         int64_t M, int32_t N, int32_t K, int32_t lda, int32_t ldd, int32_t act, int32_t out_cols,
                         int64_t out_stride, void *stream);

/* The K >= 256 engine (csrc/isg_gemm_h3p.hip): the Linears of the question encoder / decoder
 * (ISubGVQA/models/question_encoder.py:20-38, question_decoder.py:25-71: nn.TransformerEncoder/DecoderLayer's in_proj,
 * out_proj, linear1, linear2 -- cuBLAS GEMMs in the reference) and the C = 300 projections of MGAT
 * (models/mgat_v2_conv.py:177-181,259; models/mgat.py:156), same fp16 three-product arithmetic as isg_linear_f16x3, with
 * BOTH operands handed over pre-split as "planes32": row r, k-tile kt (32 columns) = one 128-byte line [hi 32 | mid 32]
 * fp16 of the row scaled by its own power of two, rows ceil(K / 32) lines long, zero padded; inv_scale[r] = 1 / scale.
 * The scale may come from any BOUND of the row's largest magnitude (the split stays exact to 2^-24 of the bound).
 * isg_planes32_elems: uint16 elements of a planes32 image of `rows` rows of K columns. */
int64_t isg_planes32_elems(int64_t rows, int32_t K);
/* fp32 rows [M, K] (stride lda) -> planes32 + inv_scale[M] (exact row maxima).  K % 4 == 0, lda % 4 == 0, 16-byte aligned. */
int isg_split_planes32(const float *a, int64_t M, int32_t K, int32_t lda, uint16_t *planes, float *inv_scale, void *stream);
/* gelu(x * instr[batch]) (ISubGVQA/models/mgat_v2_conv.py:156-157) as planes32 + inv_scale[N] -- the operand of the lin_l | lin_r
 * projection on isg_linear_h3p -- and as fp32 rows [N, C] where `rows` is not NULL (a masked layer's node gate reads them).
 * x fp32 [N, C] contiguous, instr fp32 [B, C], batch int64 [N].  4 | C, 16-byte aligned rows. */
int isg_instr_gate_planes32(const float *x, const float *instr, const int64_t *batch, float *rows, uint16_t *planes,
                            float *inv_scale, int64_t N, int32_t C, void *stream);
/* d = act(a . w^T + bias), act 0 none / 1 exact GELU / 2 ReLU.  Output, exactly one of:
 *   d        fp32 [M, ldd]                                            (d_planes = d_inv = d_bound = NULL)
 *   d_planes planes32 of the result (columns [N, roundup32(N)) written as zeros: the next Linear's k padding) + d_inv[M],
 *            scaled by the power of two of the bound
 *            |d[m, :]| < a_inv[m] * d_bound[0] + d_bound[1],  d_bound = {2^14 * max_n ||w_n||_1, max |bias|} on the device
 *            -- the next Linear's `a_planes` / `a_inv` with no pass in between (linear1 -> linear2).
 * SEGMENTED a (a_inv_first != NULL, k_split > 0 a multiple of 32): columns [0, k_split) of a row are scaled by a_inv_first[m],
 *   columns [k_split, K) by a_inv[m] -- what isg_gatv2_mp_fwd_planes writes (half rows under their own scales, each padded to
 *   whole 32-column lines; w laid out to match, zero columns under the padding).  act = 1 and K >= 512, k_split >= 512 only.
 * ISG_EUNSUPPORTED: N % 4 != 0, ldd % 4 != 0, a misaligned pointer, an operand of 2 GB or more. */
int isg_linear_h3p(const uint16_t *a_planes, const float *a_inv, const uint16_t *w_planes, const float *w_inv,
                   const float *bias, float *d, uint16_t *d_planes, float *d_inv, const float *d_bound, int64_t M,
                   int32_t N, int32_t K, int32_t ldd, int32_t act, const float *a_inv_first, int32_t k_split, void *stream);
/* The cache policy of a LARGE (>= 128 MB) fp32 result's stores in isg_linear_h3p, for the rest of the process: 0 plain, 1 nt,
 * 2 sc0 sc1 nt (write-through, streaming), -1 (the default) the built-in choice: nt at K >= 512, plain below.  Results do not
 * depend on it; the kernel's own time does, by box (profiles/r04_ag_h3p_store_policy.txt), the full model's does not.
 * ISG_EINVAL on any other value. */
int isg_linear_h3p_store_policy(int32_t policy);

/* ---------------------------------------------------------------------------------------------
 * Scene-graph encoder (ISubGVQA/models/scene_graph_encoder.py:108-143) without its concatenations
 * ------------------------------------------------------------------------------------------- */

/* out[e,:] = act( A[ia[e],:] + B[ib[e],:] + sign[e] * T[it[e],:] + D[e,:] + bias ),  e < E, C columns, fp32.
 * Replaces Linear(cat([x[row], x[col], emb])) of EdgeModel (:119-120) and Linear(cat([x[row], e'])) of NodeModel
 * (:139-140): the node parts are projected once per NODE (A, B: rows gathered by int64 index), the edge-token part is a
 * row of a projected [vocabulary, C] table (T, with the added_sym_edge sign, :80), D is a per-edge dense term.  B / T /
 * D / sign / bias may be NULL.  lda / ldb / ldt / ldd: row strides in floats (multiples of 4, so column slices of a wider
 * projection are valid operands); every pointer 16-byte aligned; act 0 = none, 1 = exact GELU.
 * planes / planes_inv (both or neither): the rows ALSO (or, with out = NULL, ONLY) as the planes32 operand of isg_linear_h3p
 * (uint16 [isg_planes32_elems(E, C)], fp32 [E]) -- the Linear that reads them (edge_mlp.2, node_mlp_1.2) needs no
 * isg_split_planes32 pass. */
int isg_gather_add(const float *A, const int64_t *ia, int32_t lda, const float *B, const int64_t *ib, int32_t ldb,
                   const float *T, const int64_t *it, const float *sign, int32_t ldt, const float *D, int32_t ldd,
                   const float *bias, float *out, int64_t E, int32_t C, int32_t act, uint16_t *planes, float *planes_inv,
                   void *stream);

/* ---------------------------------------------------------------------------------------------
 * Question encoder / program decoder attention (ISubGVQA/models/question_encoder.py:20-38, question_decoder.py:25-71)
 * ------------------------------------------------------------------------------------------- */

/* out = softmax(Q K^T / sqrt(hd) + key_bias) V per (batch item, head), for the short sequences of this model (Tk <= 128,
 * hd <= 64): what nn.MultiheadAttention computes inside nn.TransformerEncoderLayer / DecoderLayer.  Rows follow torch's
 * [T, B, D] layout (row t * B + b), heads are consecutive hd-column blocks; q / k / v may be column slices of one fused
 * projection (ldq / ldk / ldv: row strides in floats).  key_bias fp32 [B, Tk] or NULL is ADDED to the scores -- the
 * reference passes the HF attention mask as a FLOAT src_key_padding_mask (question_encoder.py:35-37: +1 on real tokens,
 * padding is attended).  out fp32 rows of H * hd columns, stride ldo.
 * rowmax fp32 [Tq * B, H] or NULL: the largest |out| of every (row, head) -- a_rowmax (P = H) of the out_proj Linear.
 * planes / planes_inv (both or neither; with them `out` may be NULL and rowmax must be): the result ALSO / ONLY as the planes32
 * operand of isg_linear_h3p (uint16 [isg_planes32_elems(Tq * B, H * hd)], fp32 [Tq * B]) -- one workgroup per batch item then
 * walks all heads and assembles the item's rows in LDS (Tq * H * hd more floats: ISG_EUNSUPPORTED beyond 64 KB in all), and
 * out_proj needs no isg_split_planes32 pass. */
int isg_mha_small(const float *q, int32_t ldq, const float *k, int32_t ldk, const float *v, int32_t ldv,
                  const float *key_bias, float *out, int32_t ldo, float *rowmax, int64_t B, int32_t H, int32_t hd,
                  int32_t Tq, int32_t Tk, uint16_t *planes, float *planes_inv, void *stream);

/* out = LayerNorm(x + r) over the last dimension, r optional (NULL): the post-norm step of nn.TransformerEncoderLayer /
 * nn.TransformerDecoderLayer (question_encoder.py:20-38, question_decoder.py:25-71) with the residual add folded in;
 * torch.nn.LayerNorm's arithmetic ((v - mean) * rstd * gamma + beta, biased variance, fp32; beta may be NULL).  rowmax
 * fp32 [M] or NULL: max |out| per row, the a_rowmax (P = 1) of the Linear that reads `out`.  planes / planes_inv (both or
 * neither): `out` also as the planes32 operand of isg_linear_h3p (uint16 [M * D * 2], fp32 [M]; 32 | D), so that the
 * Linears which read it need no isg_split_planes32 pass.  4 | D, D <= 2048, 16-byte aligned rows (ISG_EUNSUPPORTED otherwise). */
int isg_add_layernorm(const float *x, int32_t ldx, const float *r, int32_t ldr, const float *gamma, const float *beta,
                      float eps, float *out, int32_t ldo, float *rowmax, int64_t M, int32_t D, uint16_t *planes,
                      float *planes_inv, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ISG_H */
