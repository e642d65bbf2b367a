"""SIMPLE sampler: k-hot sample with exact-marginal gradients.

Reference behaviour: EdgeSIMPLEBatched, ISubGVQA/sampling/methods/simple_scheme.py:23-191, policy 'edge_candid' (the
only one ISubGVQA constructs, masking.py:110-119).  forward(scores [B, Nmax, 1], train) -> (mask [1, B, Nmax, 1],
marginals [B, Nmax, 1]) with mask = (sample - marginals).detach() + marginals.  One launch of isg_simple_topk; the
gradient of the marginals comes from autograd over sampling/methods/simple.py::log_marginals, evaluated on the device in
the backward pass only.

Extension over the reference signature: ``uniform`` (the explicit [B, n] torch.rand draw behind the Gumbel keys, for
parity runs) and ``seed`` (in-kernel Philox stream).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from ... import ops


class EdgeSIMPLEBatched(torch.nn.Module):
    def __init__(self, k, device, policy, val_ensemble=1, train_ensemble=1, logits_activation=None):
        super().__init__()
        if policy != "edge_candid":
            raise NotImplementedError(f"policy {policy!r}: ISubGVQA only uses 'edge_candid'")
        if logits_activation not in (None, "None"):
            raise NotImplementedError("logits_activation is never set by ISubGVQA (masking.py:110-119)")
        assert val_ensemble > 0 and train_ensemble > 0
        self.k, self.device, self.policy = k, device, policy
        self.val_ensemble, self.train_ensemble = val_ensemble, train_ensemble
        self.logits_activation = logits_activation
        self.adj = None

    def forward(self, scores: Tensor, train: bool = True, uniform: Optional[Tensor] = None, seed: Optional[int] = None):
        times = self.train_ensemble if train else self.val_ensemble
        if times != 1:
            raise NotImplementedError("ensembles > 1 are never used by ISubGVQA (masking.py:110-119)")
        B, nmax, ens = scores.shape
        if ens != 1:
            raise NotImplementedError("ensemble dimension must be 1")
        n = 2 ** math.ceil(math.log2(nmax)) if nmax > 1 else 1
        if uniform is None and seed is None:
            uniform = torch.rand(B, n, device=scores.device)
        mask, marg = ops.simple_topk(scores.reshape(B, nmax).contiguous(), int(self.k), uniform=uniform,
                                     seed=0 if seed is None else seed, return_marginals=True)
        return mask.view(1, B, nmax, 1), marg.view(B, nmax, 1)
