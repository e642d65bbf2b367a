"""Stack of masked GATv2 layers with head projection, instruction attention, GraphNorm and residual.

Reference behaviour: MGAT, ISubGVQA/models/mgat.py:8-184.  Per layer i:
  conv_i (MaskingGATv2Conv)  ->  x_proj_i (Linear-GELU-Linear-GELU)  ->
  scatter_scaled_dot_product_attention(ins_i, .)  ->  GraphNorm_i  ->  + h  [-> * mask]
The last four steps after x_proj are one kernel (isg_instr_attn_graphnorm_residual).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from .. import ops
from .layers import GraphNorm
from .mgat_v2_conv import MaskingGATv2Conv


class MGAT(torch.nn.Module):
    def __init__(self, channels, num_ins, dropout=0.0, heads=4, use_instr=False, masking_thresholds=None,
                 use_topk: bool = False, interpretable_mode: bool = True, concat_instr: bool = False,
                 use_all_instrs: bool = False, use_global_mask: bool = False, node_classification: bool = False,
                 sampler_type: str = None, sample_k: int = None, nb_samples: int = 1, alpha=1.0, beta=10.0, tau=1.0):
        super().__init__()
        if not use_instr:
            raise NotImplementedError("use_instr=False leaves the reference without its layer lists (mgat.py:46-64)")
        if num_ins > 4:
            raise ValueError("the reference supports at most 4 instruction layers (mgat.py:47-53)")
        # options the forward pass consults
        self.heads, self.use_instr, self.use_topk, self.dropout = heads, use_instr, use_topk, dropout
        self.masking_thresholds, self.use_global_mask = masking_thresholds, use_global_mask
        self.interpretable_mode, self.use_all_instrs = interpretable_mode, use_all_instrs
        self.node_classification = node_classification
        width_in = 2 * channels if concat_instr else channels
        self.in_channels = width_in
        wide, mid = heads * channels, channels * int(heads / 2)        # H*C -> C*H/2 -> C   (mgat.py:66-91)

        sampler_opts = dict(sampler_type=sampler_type, sample_k=sample_k, nb_samples=nb_samples, alpha=alpha, beta=beta,
                            tau=tau)

        def conv_layer(threshold):
            return MaskingGATv2Conv(width_in, channels, heads=heads, edge_dim=channels, add_self_loops=False,
                                    masking_threshold=threshold, use_instr=True, use_topk=use_topk,
                                    concat_instr=concat_instr, use_all_instrs=use_all_instrs, **sampler_opts)

        def head_projection():
            gelu = torch.nn.GELU
            return torch.nn.Sequential(torch.nn.Linear(wide, mid), gelu(), torch.nn.Linear(mid, channels), gelu())

        layers = range(num_ins)
        self.convs = torch.nn.ModuleList(conv_layer(masking_thresholds[i]) for i in layers)
        self.x_proj = torch.nn.ModuleList(head_projection() for _ in layers)
        self.bns = torch.nn.ModuleList(GraphNorm(channels) for _ in layers)
        # built by the reference and never called in forward (mgat.py:98-102): present only for checkpoint keys
        self.node_logits = torch.nn.Sequential(torch.nn.Linear(channels, 512), torch.nn.GELU(),
                                               torch.nn.Linear(512, 2577))

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    def _x_proj_reads_planes(self, i: int, rows: int, explainer: bool) -> bool:
        """x_proj[i] = Linear, GELU(exact), ... with its first Linear on the planes32 engine: the convolution may hand its result
        over as segmented planes32 (ops.gatv2_mp want_planes) instead of fp32 rows."""
        seq = self.x_proj[i]
        if explainer or torch.is_grad_enabled() or not isinstance(seq, torch.nn.Sequential) or len(seq) < 2:
            return False
        lin, act = seq[0], seq[1]
        return (isinstance(lin, torch.nn.Linear) and isinstance(act, torch.nn.GELU) and act.approximate == "none" and
                ops.MP_PLANES and ops.h3p_supported(rows, lin.weight.size(0), lin.weight.size(1)))

    def on_tiles(self, plan, in_channels: int, edge_attr) -> bool:
        """Will forward() run its convolutions on the graph-tile kernels for this batch?  (Then a model may send the graphs
        beyond a tile through ops.run_split instead of having every layer fill their rows.)"""
        if plan is None or edge_attr is None or edge_attr.dim() != 2 or torch.is_grad_enabled():
            return False
        return self.convs[0].dispatch(plan, in_channels, edge_attr.float()) in ("layer_conv", "tile_conv")

    def forward(self, x, edge_index, instr_vectors, global_language_feats, edge_attr, batch, return_masks=False,
                explainer=False, explainer_stage=False, expl_bypass_x=False, plan: Optional[ops.GraphPlan] = None,
                noises: Optional[Dict[int, Tensor]] = None, seed: Optional[int] = None,
                return_attention: bool = False, gate_feats: Optional[Tensor] = None):
        # gate_feats [B, C]: the rows the masked layers' node gates read, given per graph (ops.run_split's sub-batch); default:
        # global_language_feats under the reference's double index (masking.py:151-155, quirk Q3)
        if plan is None:
            plan = ops.GraphPlan.build(batch, edge_index, num_graphs=global_language_feats.size(0))
        h = x.float().contiguous()
        edge_attr = edge_attr.float().contiguous()
        glf = global_language_feats.contiguous()
        mask = None
        global_mask = None
        edge_attns = []
        # every layer projects the SAME edge features (mgat.py:144-148): one launch splits each 64-row panel of
        # edge_attr into its bf16 planes once and writes one dense [E, H*C] tensor per layer (isg_linear_panel_multi)
        e_projs = None
        fdt = self.convs[0].rows_dtype(plan)
        fused = (edge_attr.dim() == 2 and not torch.is_grad_enabled()
                 and (fdt == torch.float32 or (fdt == torch.float16 and edge_attr.size(1) >= 128))       # MaskingGATv2Conv.dispatch
                 and ops.fused_logits_supported(plan, self.heads, self.convs[0].out_channels, edge_attr.size(1)))
        if (not fused and not torch.is_grad_enabled() and all(c.lin_edge is not None for c in self.convs)
                and edge_attr.dim() == 2):
            e_projs = ops.linear_multi(edge_attr, [c.lin_edge.weight for c in self.convs],
                                       out_dtype=fdt)
        L = len(self.convs)
        wide = self.heads * self.convs[0].out_channels
        x_gated = x_planes = None     # gelu(h * ins_i[batch]) when the previous layer's fused tail has written it: fp32 rows, planes
        for i in range(L):
            ins = instr_vectors[i].contiguous()
            if explainer:
                h = expl_bypass_x if (explainer_stage - 1) == i else h                   # :140-141
                x_gated = x_planes = None
            conv_res, mask, edge_att = self.convs[i](
                x=h, edge_index=edge_index, edge_attr=edge_attr, instruction=ins, batch=batch,
                return_masks=return_masks, return_attention_weights=True, imle_att=glf if gate_feats is None else gate_feats.contiguous(),
                gate_rows_given=gate_feats is not None, all_instrs=instr_vectors,
                plan=plan, noise=None if noises is None else noises.get(i),
                seed=None if seed is None else seed + i,
                e_proj=None if e_projs is None else e_projs[i], x_gated=x_gated, x_planes=x_planes,
                out_planes=self._x_proj_reads_planes(i, h.size(0), explainer))                           # :144-154
            x_gated = x_planes = None
            if return_attention:
                edge_attns.append(edge_att)
            tail_mask = None
            if self.use_global_mask:                                                     # :161-162,174-175
                global_mask = mask if global_mask is None else mask * global_mask
                tail_mask = global_mask
            elif self.interpretable_mode and mask is not None:                           # :176-177
                tail_mask = mask
            bn = self.bns[i]
            if (isinstance(conv_res, Tensor) and conv_res.dtype == torch.float32 and not explainer
                    and ops.dense_tail_supported(plan, self.x_proj[i], wide, self.convs[0].out_channels)):
                # x_proj + instruction attention + GraphNorm + residual (+ mask) + the NEXT layer's instruction gate: one
                # launch on graph-aligned row tiles (csrc/isg_layer_tile.hip); :156-177 and mgat_v2_conv.py:156-157
                nxt = instr_vectors[i + 1].contiguous() if i + 1 < L and self.convs[i + 1].use_instr else None
                # the next layer's gated input: as planes when it runs as isg_gatv2_layer_conv, as fp32 rows when anything else
                # reads it (a masked layer's node gate, the un-fused convolution)
                want_planes = want_rows = False
                if nxt is not None:
                    cn = self.convs[i + 1]
                    want_planes = cn.layer_conv_ready(plan, h.size(1), edge_attr,
                                                      None if e_projs is None else e_projs[i + 1])
                    want_rows = cn.needs_rows(plan, h.size(1), edge_attr, None if e_projs is None else e_projs[i + 1], glf)
                res = ops.mgat_dense_tail(conv_res, self.x_proj[i], ins, h, plan, bn.weight, bn.bias, bn.mean_scale, bn.eps,
                                          node_mask=tail_mask, ins_next=nxt, want_rows=want_rows, want_planes=want_planes)
                if res is not None:
                    h, x_gated, x_planes = res
                    continue
            conv_res = ops.mlp(self.x_proj[i], conv_res)                                 # :156 (Linear+GELU fused)
            h = ops.mgat_layer_tail(ins, conv_res.contiguous(), h, plan, bn.weight, bn.bias, bn.mean_scale, bn.eps,
                                    node_mask=tail_mask)                                 # :168-177
        if return_attention:
            return h, mask, [], [], edge_attns
        return h, mask, [], []
