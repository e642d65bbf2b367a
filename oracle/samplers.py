"""Discrete top-k node-mask samplers, restated on torch CPU ops (forward only).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

All three samplers take the dense-padded gate ``scores[B, Nmax, 1]`` produced by
``to_dense_batch`` (masking.py:162) and an EXPLICIT noise tensor laid out as the
reference draws it, so that a run is reproducible independent of the RNG.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

F32_TINY = float(np.finfo(np.float32).tiny)   # gumbel_scheme.py:9  EPSILON
F32_EPS = float(torch.finfo(torch.float32).eps)


def uniform_to_gumbel(u01: Tensor, loc: float = 0.0, scale: float = 1.0) -> Tensor:
    """torch.distributions.Gumbel(loc, scale).sample() given the raw torch.rand draw.

    Gumbel is TransformedDistribution(Uniform(tiny, 1-eps), [log, *-1, log, loc - scale*x]);
    Uniform.rsample = low + rand*(high-low).  (gumbel_scheme.py:65-69, noise.py:86-89.)
    """
    low = torch.tensor(F32_TINY, dtype=torch.float32)
    high = torch.tensor(1.0 - F32_EPS, dtype=torch.float32)
    u = low + u01 * (high - low)
    return loc - scale * torch.log(-torch.log(u))


def gumbel_relaxed_topk(scores: Tensor, k: int, noise: Tensor, tau: float = 0.1,
                        hard: bool = True):
    """GumbelSampler.forward, policy 'edge_candid', ensemble 1 (gumbel_scheme.py:55-58,63-107).

    scores [B, Nmax, 1]; noise [B, Nmax] = Gumbel(0,1) sample.  Returns
    (new_mask [1, B, Nmax, 1], khot [B, Nmax], ind [B, local_k]).
    Noise is added in eval mode as well (masking.py:175 calls forward(train=False)).
    """
    B, Nmax, ens = scores.shape
    assert ens == 1
    flat = scores.permute(0, 2, 1).reshape(B * ens, Nmax)        # :57
    local_k = min(k, Nmax)                                       # :58
    flat = flat + noise                                          # :70
    khot = torch.zeros_like(flat)
    onehot = torch.zeros_like(flat)
    for _ in range(local_k):                                     # :75-81
        khot_mask = torch.max(1.0 - onehot, torch.tensor([F32_TINY]))
        flat = flat + torch.log(khot_mask)
        onehot = torch.softmax(flat / tau, dim=1)
        khot = khot + onehot
    ind = None
    if hard:                                                     # :83-88
        khot_hard = torch.zeros_like(khot)
        _, ind = torch.topk(khot, local_k, dim=1)
        khot_hard = khot_hard.scatter_(1, ind, 1)
        res = khot_hard - khot + khot
    else:
        res = khot
    new_mask = res.reshape(1, B, ens, Nmax).permute(0, 1, 3, 2)   # :102-104
    return new_mask, khot, ind


def threshold_topk(scores: Tensor, k: int) -> Tensor:
    """select_from_edge_candidates (deterministic_scheme.py:36-43): all ties at the k-th value kept."""
    B, Nmax, ens = scores.shape
    if k >= Nmax:
        return torch.ones_like(scores)
    thresh = torch.topk(scores, k, dim=1, largest=True, sorted=True).values[:, -1, :][:, None, :]
    return (scores >= thresh).to(torch.float)


def imle_eval(scores: Tensor, k: int, noise: Tensor | None = None,
              input_noise_temperature: float = 0.0):
    """imle wrapper forward, nb_samples=1 (wrapper.py:75-121) over IMLEScheme (imle_scheme.py:17-29).

    Eval sampler from get_imle_samplers has input_noise_temperature=0.0 (masking.py:238):
    the noise is drawn and multiplied by zero.  noise [B, 1, Nmax, 1].
    Returns res [1, B, Nmax, 1] (masking.py:170 takes output[0].squeeze(0)[mask]).
    """
    B, Nmax, ens = scores.shape
    if noise is None:
        noise = torch.zeros(B, 1, Nmax, ens)
    pert = scores[:, None, ...] + noise * input_noise_temperature   # :93-100
    out = threshold_topk(pert.view(B, Nmax, ens), k)                # :103-110
    return out.view(B, 1, Nmax, ens).permute(1, 0, 2, 3)            # :118


def aimle_eval(scores: Tensor, k: int, noise: Tensor, theta_noise_temperature: float = 1.0):
    """aimle wrapper forward, nb_samples=1 (aimle.py:83-138).

    Eval sampler from get_aimle_samplers: noise ~ Gumbel(0, 0.3) of shape [B,1,Nmax,1],
    theta_noise_temperature = tau (masking.py:262,275) -> stochastic in eval.
    Returns z [B, Nmax, 1] (masking.py:172 takes output[mask]).
    """
    B, Nmax, ens = scores.shape
    eps = noise * theta_noise_temperature                           # :109
    pert = scores.view(B, 1, -1).repeat(1, 1, 1).view(B, 1, Nmax, ens) + eps   # :112-117
    return threshold_topk(pert.view(B, Nmax, ens), k)               # :120-138
