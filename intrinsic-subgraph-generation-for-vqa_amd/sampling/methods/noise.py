"""Gumbel perturbation noise for the perturb-and-MAP samplers.

Reference behaviour: ISubGVQA/sampling/methods/noise.py:73-89 builds torch.distributions.Gumbel and
samples on the CPU before moving to the device.  Here the uniform draw comes from the device
generator (reproducible under torch.manual_seed) and goes through the same transform chain
Uniform(tiny, 1-eps) -> log -> negate -> log -> loc - scale * x.
"""
from __future__ import annotations

import torch
from torch import Tensor

_F32 = torch.finfo(torch.float32)


def gumbel_from_uniform(u01: Tensor, loc: float = 0.0, scale: float = 1.0) -> Tensor:
    lo, hi = _F32.tiny, 1.0 - _F32.eps
    u = lo + u01 * (hi - lo)
    return loc - scale * torch.log(-torch.log(u))


class GumbelDistribution:
    def __init__(self, loc: float = 0.0, scale: float = 1.0, device="cpu"):
        self.loc = loc
        self.scale = scale
        self.device = device

    def sample(self, shape) -> Tensor:
        return gumbel_from_uniform(torch.rand(tuple(shape), device=self.device), self.loc, self.scale)
