"""Threshold top-k MAP solver used by I-MLE / AIMLE at inference, and the scheme object that
carries its configuration.

Reference behaviour: ISubGVQA/sampling/methods/deterministic_scheme.py:36-43 (policy 'edge_candid')
and ISubGVQA/sampling/methods/imle_scheme.py:8-29.  The arithmetic is isg_topk_threshold
(csrc/isg_sampler.hip): one wave per row, k rounds of wave-max to find the k-th largest value,
then `v >= thresh` -- so every tie at the k-th value is selected and k >= Nmax selects everything.
"""
from __future__ import annotations

import torch

from ... import ops

_UNUSED_POLICIES = ("global_directed", "global_undirected")   # never chosen by ISubGVQA (masking.py:218,252)


def select_from_edge_candidates(scores: torch.Tensor, k: int) -> torch.Tensor:
    """scores [B, Nmax, 1] -> float {0,1} mask [B, Nmax, 1]."""
    if scores.dim() != 3 or scores.size(2) != 1:
        raise NotImplementedError(f"expected scores [B, Nmax, 1] (ensemble of one), got {tuple(scores.shape)}")
    B, nmax, _ = scores.shape
    dense = scores.detach().reshape(B, nmax).contiguous()
    return ops.topk_threshold(dense, int(k)).view(B, nmax, 1)


class IMLEScheme:
    """Solver handle passed to the imle / aimle decorators (reference: imle_scheme.py)."""

    def __init__(self, imle_sample_policy, sample_k, train_ensemble, val_ensemble):
        if imle_sample_policy in _UNUSED_POLICIES:
            raise NotImplementedError(f"policy {imle_sample_policy!r} is outside the ISubGVQA path")
        if imle_sample_policy != "edge_candid":
            raise NotImplementedError(imle_sample_policy)
        self.policy = imle_sample_policy
        self.k = sample_k
        self.adj = None
        self.train_ensemble = train_ensemble
        self.val_ensemble = val_ensemble

    def torch_sample_scheme(self, logits: torch.Tensor):
        with torch.no_grad():
            return select_from_edge_candidates(logits, self.k), None
