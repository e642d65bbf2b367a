"""Node mask -> edge mask.

Reference behaviour: NodeMaskToEdgeMask.forward, ISubGVQA/sampling/node_edge_masks.py:7-10:
edge_mask = mask[src] * mask[dst].  Kernel: isg_node_to_edge_mask.  (Inside the convolution the
product is fused into the message-passing kernel and this tensor is never materialised.)
"""
from __future__ import annotations

import torch

from .. import ops


class NodeMaskToEdgeMask:
    """Keeps the reference's call form ``NodeMaskToEdgeMask.apply(mask, edge_index, n_nodes)``."""

    @staticmethod
    def apply(mask: torch.Tensor, edge_index: torch.Tensor, n_nodes=None) -> torch.Tensor:
        if torch.is_grad_enabled() and mask.requires_grad:
            raise NotImplementedError("NodeMaskToEdgeMask.backward is SURVEY §8f row 1")
        return ops.node_to_edge_mask(mask.float().contiguous(), edge_index)
