"""Small parameter holders whose state_dict keys match the PyG modules the reference uses
(SURVEY Appendix C), backed by the HIP operators where they do sparse work."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from .. import ops


def glorot_(t: Tensor) -> None:
    """torch_geometric.nn.inits.glorot: U(-a, a), a = sqrt(6 / (fan_in + fan_out))."""
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class GlorotLinear(torch.nn.Module):
    """Keys ``weight`` [out,in] and ``bias`` like torch_geometric.nn.dense.linear.Linear with
    weight_initializer='glorot' and zero bias (mgat_v2_conv.py:64-101).  y = x W^T + b."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = torch.nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, x: Tensor) -> Tensor:
        return F.linear(x, self.weight, self.bias)


class GraphNorm(torch.nn.Module):
    """PyG GraphNorm (keys weight, bias, mean_scale; eps 1e-5) on isg_graph_norm (mgat.py:93-95,171)."""

    def __init__(self, in_channels: int, eps: float = 1e-5):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = torch.nn.Parameter(torch.ones(in_channels))
        self.bias = torch.nn.Parameter(torch.zeros(in_channels))
        self.mean_scale = torch.nn.Parameter(torch.ones(in_channels))

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)
        torch.nn.init.zeros_(self.bias)
        torch.nn.init.ones_(self.mean_scale)

    def forward(self, x: Tensor, batch: Optional[Tensor] = None, batch_size: Optional[int] = None,
                plan: Optional[ops.GraphPlan] = None, fp64: bool = False) -> Tensor:
        if plan is None:
            if batch is None:
                batch = torch.zeros(x.size(0), dtype=torch.long, device=x.device)
                batch_size = 1
            plan = ops.GraphPlan.build(batch, None, num_graphs=batch_size)
        return ops.graph_norm(x.float().contiguous(), plan, self.weight, self.bias, self.mean_scale, self.eps, fp64)


class _SelectTopK(torch.nn.Module):
    def __init__(self, in_channels: int):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(1, in_channels))
        bound = 1.0 / math.sqrt(in_channels)
        torch.nn.init.uniform_(self.weight, -bound, bound)


class TopKPoolingParams(torch.nn.Module):
    """The reference constructs torch_geometric.nn.TopKPooling but never calls it (masking.py:89-90);
    only its parameter (PyG 2.6.1 key ``select.weight`` [1,C]) has to exist for strict checkpoint loads."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.select = _SelectTopK(in_channels)
