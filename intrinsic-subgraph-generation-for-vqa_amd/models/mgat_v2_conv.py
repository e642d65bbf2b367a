"""GATv2 convolution with edge features, instruction gating and node->edge masking.

Reference behaviour: MaskingGATv2Conv, ISubGVQA/models/mgat_v2_conv.py:18-285 (constructor keywords,
forward signature and return tuples are kept; ``plan``/``noise``/``seed`` are optional extras).

Device work per call:
  isg_instr_gate                 x = gelu(x * instruction[batch])                       (:156-157)
  MaskingModel (masked layers)   node gate + top-k sampler -> node mask [N,1]           (:161-168)
  three dense projections        lin_l, lin_r (C -> H*C, +bias), lin_edge (no bias)     (:177,181,259)
  isg_gatv2_mp_fwd               message + segment softmax + aggregate + bias, with the node->edge
                                 mask product of NodeMaskToEdgeMask fused in            (:169-171,215-232,243-279)
The reference stashes alpha on the module between message() and forward() (:223-224,274); here it
is a local, so the module is re-entrant.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import Tensor
from torch.nn import Parameter

from .. import ops
from ..sampling.node_edge_masks import NodeMaskToEdgeMask
from .layers import GlorotLinear, glorot_
from .masking import MaskingModel


class MaskingGATv2Conv(torch.nn.Module):
    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int, heads: int = 1,
                 concat: bool = True, negative_slope: float = 0.2, dropout: float = 0.0, add_self_loops: bool = True,
                 edge_dim: Optional[int] = None, fill_value="mean", bias: bool = True, share_weights: bool = False,
                 masking_threshold=None, use_instr: bool = False, use_topk: bool = False, concat_instr: bool = False,
                 use_all_instrs: bool = False, sampler_type: str = None, sample_k: int = None, nb_samples: int = 1,
                 alpha=1.0, beta=10.0, tau=1.0, **kwargs):
        super().__init__()
        if not isinstance(in_channels, int):
            raise NotImplementedError("bipartite in_channels=(src, dst) is never used by ISubGVQA (mgat.py:58)")
        if add_self_loops:
            raise NotImplementedError("add_self_loops=True: MGAT passes False (mgat.py:63); scene graphs carry their "
                                      "self-loops already (datasets/scene_graph.py:309-343)")
        if not concat:
            raise NotImplementedError("concat=False (head averaging) is never used by ISubGVQA")
        if concat_instr or use_all_instrs:
            raise NotImplementedError("concat_instr / use_all_instrs are off by default (arg_parser.py:102,108) and "
                                      "outside this path")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops, self.edge_dim, self.fill_value = add_self_loops, edge_dim, fill_value
        self.share_weights, self.use_instr = share_weights, use_instr
        self.concat_instr, self.use_all_instrs = concat_instr, use_all_instrs

        self.lin_l = GlorotLinear(in_channels, heads * out_channels, bias=bias)
        self.lin_r = self.lin_l if share_weights else GlorotLinear(in_channels, heads * out_channels, bias=bias)
        self.att = Parameter(torch.empty(1, heads, out_channels))
        self.lin_edge = GlorotLinear(edge_dim, heads * out_channels, bias=False) if edge_dim is not None else None
        if bias:
            self.bias = Parameter(torch.empty(heads * out_channels))
        else:
            self.register_parameter("bias", None)
        self.mask = MaskingModel(in_channels, out_channels, masking_threshold, use_topk=use_topk,
                                 sampler_type=sampler_type, sample_k=sample_k, nb_samples=nb_samples, alpha=alpha,
                                 beta=beta, tau=tau)
        self.masking = NodeMaskToEdgeMask.apply
        # storage type of the projected rows x_l / x_r / e_proj and of the aggregated output (fp32 arithmetic either way):
        # torch.float16 is BASELINE configs[4] ("fp16 features / fp32 accumulate"), inference only
        self.feature_dtype = torch.float32
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()
        if self.lin_edge is not None:
            self.lin_edge.reset_parameters()
        glorot_(self.att)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def rows_dtype(self, plan) -> torch.dtype:
        """The storage type of x_l / x_r / e_proj / out for THIS batch: feature_dtype -- unless it is half and a graph of the batch lies
        beyond the per-graph kernel's 256-node / 1 024-slot tables (GK_NCAP_L / GK_ECAP_L, csrc/isg_mp_graph.hip), the only kernels
        that read half rows: such a batch keeps fp32 rows (more precise than the half rows it was asked for, never less; within the
        1e-3 the fp16 mode is held to) instead of raising ISG_EUNSUPPORTED from the message-passing launch."""
        fdt = self.feature_dtype
        if fdt == torch.float16 and plan is not None and (plan.nmax > ops.GK_NCAP_L or plan.emax > ops.GK_ECAP_L):
            return torch.float32
        return fdt

    def dispatch(self, plan, in_channels: int, edge_attr, e_proj=None) -> str:
        """Which kernels run this layer's message passing -- the ONE place that decides (forward, layer_conv_ready and
        needs_rows all ask here):
          "layer_conv"  lin_l | lin_r, lin_edge, logits, softmax, aggregation as one persistent launch on graph tiles
          "tile_conv"   the same with x_l / x_r projected before it (a layer input that is not 128 wide)
          "pair"        lin_edge folded into the logits, softmax + aggregation from them (two launches, per-graph kernel; also on
                        fp16 feature rows when the edge width is one the rows kernel takes)
          "unfused"     lin_edge as a Linear (e_proj in memory) + the message-passing kernel (any width, narrow fp16 rows, training)"""
        if (e_proj is not None or edge_attr is None or self.lin_edge is None or edge_attr.dim() != 2
                or torch.is_grad_enabled() or plan is None):
            return "unfused"
        # forward, layer_conv_ready and needs_rows of this layer and MGAT's look-ahead ask ~6 times per layer and step: the answer
        # depends on the plan, the widths, the storage type and the switches only, and is kept on the plan (host time: 0.15 ms per step)
        memo = plan.memo()
        key = ("dispatch", id(self), in_channels, edge_attr.size(1), self.rows_dtype(plan), self.share_weights)
        hit = memo.get(key)
        if hit is not None and hit[0] is ops.CFG:
            return hit[1]
        how = self._dispatch(plan, in_channels, edge_attr)
        memo[key] = (ops.CFG, how)
        return how

    def _dispatch(self, plan, in_channels: int, edge_attr) -> str:
        H, C = self.heads, self.out_channels
        if not ops.fused_logits_supported(plan, H, C, edge_attr.size(1)):
            return "unfused"
        if self.rows_dtype(plan) != torch.float32:
            # fp16 feature rows (BASELINE configs[4]): the pair exists on the rows kernel (K >= 128), the tile kernels do not
            return "pair" if self.feature_dtype == torch.float16 and edge_attr.size(1) >= 128 else "unfused"
        if not self.share_weights and ops.layer_conv_supported(plan, H, C, in_channels, edge_attr.size(1)):
            return "layer_conv"
        if ops.tile_conv_supported(plan, H, C, edge_attr.size(1)):
            return "tile_conv"
        return "pair"

    def layer_conv_ready(self, plan, in_channels: int, edge_attr, e_proj=None) -> bool:
        """Will forward() run this layer as isg_gatv2_layer_conv?  (Then its gated input is wanted as ops.NodePlanes: MGAT asks
        before it lets the previous layer's fused tail write them.)"""
        return self.dispatch(plan, in_channels, edge_attr, e_proj) == "layer_conv"

    def needs_rows(self, plan, in_channels: int, edge_attr, e_proj=None, imle_att=None) -> bool:
        """Does forward() read the gated layer input as fp32 ROWS (beside, or instead of, its planes)?  Everything but the
        layer kernel does, and so does a node gate that cannot run on the planes."""
        if self.dispatch(plan, in_channels, edge_attr, e_proj) != "layer_conv":
            return True
        return self.mask.masking_threshold != 1.0 and not self.mask.planes_ready(imle_att)

    def forward(self, x: Tensor, edge_index: Tensor, batch: Tensor, edge_attr: Optional[Tensor] = None,
                instruction: Optional[Tensor] = None, imle_att: Optional[Tensor] = None,
                return_attention_weights: bool = None, return_masks: bool = None, all_instrs=None,
                plan: Optional[ops.GraphPlan] = None, noise: Optional[Tensor] = None, seed: Optional[int] = None,
                e_proj: Optional[Tensor] = None, x_gated: Optional[Tensor] = None,
                x_planes: Optional["ops.NodePlanes"] = None, out_planes: bool = False, gate_rows_given: bool = False):
        # gate_rows_given: imle_att[g] already IS the row the node gate of graph g reads (ops.run_split's sub-batch: the reference's
        # double index batch[batch[n]], quirk Q3, refers to positions in the batch the graphs were taken from)
        # out_planes (inference): the caller feeds the result to a Linear + GELU on the planes32 engine (MGAT's x_proj.0) and
        # takes it as a segmented ops.Planes32 where the message-passing kernel can write that (H = 4, the flat kernel)
        H, C = self.heads, self.out_channels
        if x.dim() != 2:
            raise ValueError("x must be [N, C]")
        if self.dropout != 0.0 and self.training:
            raise NotImplementedError("attention dropout in training mode (forward-only path)")
        if plan is None:
            plan = ops.GraphPlan.build(batch, edge_index,
                                       num_graphs=None if instruction is None else instruction.size(0))
        how = self.dispatch(plan, x.size(1), edge_attr, e_proj)
        need_rows = self.needs_rows(plan, x.size(1), edge_attr, e_proj, imle_att)
        planes = None          # gelu(x * instruction[batch]) as the planes isg_gatv2_layer_conv reads (fp32 rows only where needed)
        if (x_gated is not None or x_planes is not None) and self.use_instr:
            # gelu(x * instruction[batch]) was already written by the previous layer's fused tail (isg_mgat_dense_tail)
            x, planes = x_gated, x_planes
            if x is None and need_rows:
                raise RuntimeError("the previous layer's tail left no fp32 rows of the gated input, and this layer needs them")
        else:
            x = x.float().contiguous()
            if self.use_instr and how == "layer_conv":
                x, planes = ops.instr_gate_planes(x, instruction.contiguous(), batch, want_rows=need_rows)   # :156-157
            elif self.use_instr and not torch.is_grad_enabled() and self.rows_dtype(plan) == torch.float32 and \
                    ops.h3p_supported(x.size(0), (1 if self.share_weights else 2) * H * C, x.size(1)):
                # the projection runs on the planes32 engine: the gate writes its operand (and fp32 rows only for a node gate)
                masked = self.mask.masking_threshold != 1.0
                rows, xp = ops.instr_gate_planes32(x, instruction.contiguous(), batch, want_rows=masked)   # :156-157
                x = rows if masked else xp
            elif self.use_instr:
                x = ops.instr_gate(x, instruction.contiguous(), batch, plan=plan)        # :156-157

        mask = None
        if self.mask.masking_threshold != 1.0:                                           # :161
            mask = self.mask(x, imle_att, batch, edge_index, use_all_instrs=False, plan=plan, noise=noise,
                             seed=seed, u_is_per_graph=not gate_rows_given, x_planes=planes)             # :166-168

        def done(out, alpha):
            if isinstance(return_attention_weights, bool):
                return out, mask, (edge_index, alpha)                                    # :237
            return out, mask                                                             # :241

        fdt = self.rows_dtype(plan)
        kw = dict(bias=self.bias, node_mask=mask, negative_slope=self.negative_slope)
        if how == "layer_conv":
            # lin_l | lin_r, lin_edge, logits, softmax and aggregation as ONE persistent launch on graph-aligned tiles
            # (csrc/isg_layer_conv.hip): x_l / x_r live in LDS only                       # :177-181, :215-232, :243-279
            res = ops.gatv2_layer_conv(planes if planes is not None else x, self.lin_l, self.lin_r, edge_attr.float().contiguous(),
                                       self.lin_edge.weight, self.att, plan, H, want_rowmax=True, **kw)
            if res is not None:
                return done(*res)
            if x is None:
                raise RuntimeError("isg_gatv2_layer_conv refused a shape layer_conv_supported() accepted, and the gated input "
                                   "exists only as planes")
            how = "tile_conv"
        if self.share_weights:
            x_l = x_r = ops.linear(x, self.lin_l.weight, self.lin_l.bias, out_dtype=fdt)  # :177-179
        else:   # lin_l and lin_r share their input: one [N, 2*H*C] projection, x_l / x_r are its column halves
            x_l, x_r = ops.linear_fused(x, (self.lin_l, self.lin_r), out_dtype=fdt)      # :177,181
        if how == "tile_conv" and ops.tile_conv_supported(plan, H, C, edge_attr.size(1)):
            # edge GEMM + logits + softmax + aggregation as one launch on graph-aligned tiles (csrc/isg_layer_tile.hip)
            res = ops.gatv2_tile_conv(x_l, x_r, edge_attr.float().contiguous(), self.lin_edge.weight, self.att, plan, H,
                                      want_rowmax=True, **kw)                            # :215-232, :243-279
            if res is not None:
                return done(*res)
            how = "pair"
        if how in ("tile_conv", "pair"):
            # lin_edge folded into the logits (csrc/isg_mp_logits.hip): e_proj [E, H*C] is neither written nor read
            res = ops.gatv2_mp_edge_logits(x_l, x_r, edge_attr.float().contiguous(), self.lin_edge.weight, self.att, plan, H,
                                           want_rowmax=True, want_planes=out_planes, **kw)          # :215-232, :243-279
            if res is not None:
                return done(*res)
        if e_proj is None:
            if edge_attr is None or self.lin_edge is None:
                raise NotImplementedError("edge_attr=None: MGAT always passes edge features (mgat.py:147)")
            if edge_attr.dim() == 1:
                edge_attr = edge_attr.view(-1, 1)
            e_proj = ops.linear(edge_attr.float().contiguous(), self.lin_edge.weight, None, out_dtype=fdt)   # :259
        return done(*ops.gatv2_mp(x_l, x_r, e_proj, self.att, plan, H, want_rowmax=not torch.is_grad_enabled(),
                                  want_planes=out_planes and not torch.is_grad_enabled(), **kw))          # :215-232

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, heads={self.heads})"
