"""Relaxed Gumbel top-k sampler with a straight-through hard mask.

Reference behaviour: GumbelSampler.forward, ISubGVQA/sampling/methods/gumbel_scheme.py:26-107 with
policy 'edge_candid' (the only one ISubGVQA constructs, masking.py:121-123).  Note the reference adds
Gumbel(0,1) noise in eval mode too and selects through k rounds of a tau=0.1 softmax -- not a plain
top-k; both are reproduced by isg_topk_gumbel (csrc/isg_sampler.hip).

Extension over the reference signature: ``noise`` (an explicit [B, Nmax] Gumbel(0,1) draw, for parity
runs) and ``seed`` (in-kernel Philox stream).  With neither, the draw comes from torch's device
generator so torch.manual_seed reproduces a run.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from ... import ops
from .deterministic_scheme import select_from_edge_candidates
from .noise import gumbel_from_uniform


class GumbelSampler(torch.nn.Module):
    def __init__(self, k, train_ensemble, val_ensemble, tau=0.1, hard=True, policy=None):
        super().__init__()
        if policy != "edge_candid":
            raise NotImplementedError(f"policy {policy!r}: ISubGVQA only uses 'edge_candid'")
        if not hard:
            raise NotImplementedError("hard=False (soft k-hot output) is never used by ISubGVQA")
        self.policy, self.k, self.hard, self.tau = policy, k, hard, tau
        self.adj = None
        self.train_ensemble, self.val_ensemble = train_ensemble, val_ensemble

    def forward(self, scores: Tensor, train: bool = True, noise: Optional[Tensor] = None,
                seed: Optional[int] = None):
        repeat = self.train_ensemble if train else self.val_ensemble
        if repeat != 1:
            raise NotImplementedError("ensembles > 1 are never used by ISubGVQA (masking.py:122)")
        B, nmax, ens = scores.shape
        if ens != 1:
            raise NotImplementedError("ensemble dimension must be 1")
        dense = scores.reshape(B, nmax).contiguous()      # differentiable: straight-through backward (:83-90)
        if noise is None and seed is None:
            noise = gumbel_from_uniform(torch.rand(B, nmax, device=scores.device))
        res = ops.topk_gumbel(dense, int(self.k), float(self.tau), noise=noise, seed=0 if seed is None else seed)
        return res.view(1, B, nmax, 1), None

    @torch.no_grad()
    def validation(self, scores: Tensor):
        if self.val_ensemble != 1:
            return self.forward(scores, False)
        return select_from_edge_candidates(scores, self.k)[None], None
