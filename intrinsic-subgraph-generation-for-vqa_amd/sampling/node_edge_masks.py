"""Node mask -> edge mask.

Reference behaviour: NodeMaskToEdgeMask, ISubGVQA/sampling/node_edge_masks.py:7-19: forward
edge_mask = mask[src] * mask[dst] (isg_node_to_edge_mask); backward scatters the edge-mask gradient to the
DESTINATION node only (isg_node_to_edge_mask_bwd) -- the reference's rule, not the product rule.  (Inside the
convolution the product is fused into the message-passing kernel and this tensor is never materialised; its
backward applies the same rule.)
"""
from __future__ import annotations

import torch

from .. import ops


class NodeMaskToEdgeMask:
    """Keeps the reference's call form ``NodeMaskToEdgeMask.apply(mask, edge_index, n_nodes)``."""

    @staticmethod
    def apply(mask: torch.Tensor, edge_index: torch.Tensor, n_nodes=None) -> torch.Tensor:
        mask = mask.float().contiguous()
        if torch.is_grad_enabled() and mask.requires_grad:
            return ops.node_to_edge_mask(mask, edge_index, ops.GraphPlan.edges_only(edge_index, mask.shape[0]))
        return ops.node_to_edge_mask(mask, edge_index)
