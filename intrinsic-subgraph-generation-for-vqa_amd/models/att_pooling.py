"""Question-conditioned global attention pooling.

Reference behaviour: GlobalAttention, ISubGVQA/models/att_pooling.py:6-82:
  x' = node_nn(x) * node_mask;  gate = softmax_g(<x'_n, ques_nn(u)_g>/sqrt(C));  out_g = sum_n gate_n x'_n
The two small MLPs are dense GEMMs; the mask product, per-graph softmax (PyG form, +1e-16) and the
scatter-add are one kernel (isg_global_attn_pool).  The reference's hard-coded ``batch.cuda()``
(:71,73) has no counterpart: tensors are used where they live.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops


class GlobalAttention(torch.nn.Module):
    def __init__(self, num_node_features, num_out_features):
        super().__init__()
        channels = num_out_features
        self.gate_nn = torch.nn.Sequential(torch.nn.Linear(channels, channels), torch.nn.GELU(),
                                           torch.nn.Linear(channels, 1))                # unused in forward
        self.node_nn = torch.nn.Sequential(torch.nn.Linear(num_node_features, channels), torch.nn.GELU(),
                                           torch.nn.Linear(channels, channels))
        self.ques_nn = torch.nn.Sequential(torch.nn.Linear(channels, channels), torch.nn.GELU(),
                                           torch.nn.Linear(channels, channels))

    def reset_parameters(self):
        for seq in (self.gate_nn, self.node_nn, self.ques_nn):
            for m in seq:
                if hasattr(m, "reset_parameters"):
                    m.reset_parameters()

    def forward(self, x, u, batch, size=None, return_mask=False, node_mask=None,
                plan: Optional[ops.GraphPlan] = None):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        if plan is None:
            plan = ops.GraphPlan.build(batch, None, num_graphs=u.size(0) if size is None else size)
        q = ops.mlp(self.ques_nn, u)                                                     # :66
        res = None
        if x.dtype == torch.float32 and x.dim() == 2 and ops.readout_tile_supported(plan, self.node_nn, x.size(1)):
            # node_nn, mask, per-graph softmax and pooled sum as one launch on graph-aligned tiles (csrc/isg_layer_conv.hip)
            res = ops.readout_tile(x.contiguous(), self.node_nn, q.contiguous(), plan, node_mask)     # :62-73
        if res is not None:
            out, gate = res
        else:
            xn = ops.mlp(self.node_nn, x)                                                # :62
            out, gate = ops.global_attn_pool(xn.contiguous(), q.contiguous(), plan, node_mask)   # :63-73
        if return_mask:
            return out, gate
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}(gate_nn={self.gate_nn}, node_nn={self.node_nn}, ques_nn={self.ques_nn})"
