"""Instruction -> node scatter-softmax attention.

Reference behaviour: ISubGVQA/utils/scatter_scaled_dot_product.py:6-15 -- per graph,
softmax_n(<query_g, key_n>/sqrt(C)) (torch_scatter.scatter_softmax, no epsilon) scales value_n.
Kernel: isg_scatter_attention (one workgroup per graph).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops


def scatter_scaled_dot_product_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                                         batch: torch.Tensor, plan: Optional[ops.GraphPlan] = None) -> torch.Tensor:
    if plan is None:
        plan = ops.GraphPlan.build(batch, None, num_graphs=query.size(0))
    return ops.scatter_attention(query.contiguous(), key.contiguous(), plan, value.contiguous())
