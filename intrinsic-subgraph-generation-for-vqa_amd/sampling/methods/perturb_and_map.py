"""The I-MLE and AIMLE perturb-and-MAP wrappers.

Backward: I-MLE solves MAP(alpha*theta - beta*dy + tau*eps) once more and returns z - z'
(wrapper.py:124-172); AIMLE solves the two symmetric targets alpha*theta -/+ lambda*dy and adapts lambda's beta
(aimle.py:141-243, target_aimle.py:88-162).  Both are autograd.imle_topk / aimle_topk: the same isg_topk_threshold
kernel on the target scores with the forward's noise.

Reference behaviour of the forward:
  * I-MLE  ISubGVQA/sampling/methods/wrapper.py:75-121: noise [B, S, ...] is drawn, scaled by
    input_noise_temperature, added to the input, the solver is run on [B*S, ...] and the result is
    returned as ([S, B, ...], aux).
  * AIMLE  ISubGVQA/sampling/methods/aimle.py:83-138: same perturbation with theta_noise_temperature;
    returns z [B*S, ...].
Both end in the threshold top-k, so with S = 1 and an 'edge_candid' scheme the perturbation and the
solve are one kernel launch (isg_topk_threshold with noise and noise_scale).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor

from ... import ops


class PerturbAndMAP:
    def __init__(self, function: Callable, noise_distribution, nb_samples: int, noise_temperature: float,
                 kind: str, target_distribution=None, target_noise_temperature: float = 1.0):
        self.function = function
        self.noise_distribution = noise_distribution
        self.nb_samples = int(nb_samples)
        self.noise_temperature = float(noise_temperature)
        self.kind = kind                                   # "imle" | "aimle"
        self.target_noise_temperature = float(target_noise_temperature)
        self.__name__ = getattr(function, "__name__", "perturb_and_map")
        # backward only.  I-MLE: (alpha, beta); AIMLE: the stateful adaptive target (one per sampler, like the reference)
        from .target import TargetDistribution
        from .target_aimle import AdaptiveTargetDistribution
        if isinstance(target_distribution, tuple):        # ("imle", alpha, beta) / ("aimle", alpha, beta0) shorthand
            _, a, b = target_distribution
            target_distribution = (TargetDistribution(alpha=a, beta=b) if kind == "imle"
                                   else AdaptiveTargetDistribution(initial_alpha=a, initial_beta=b))
        if target_distribution is None and kind == "imle":
            target_distribution = TargetDistribution(alpha=1.0, beta=1.0)               # wrapper.py:57-58
        self.target_distribution = target_distribution

    def _scheme_k(self) -> Optional[int]:
        """k of the wrapped IMLEScheme when the solver is the stock threshold top-k, else None."""
        return getattr(self.function, "_isg_threshold_k", None)

    def differentiable(self, scores: Tensor, plan, noise: Optional[Tensor], seed: int = 0) -> Tensor:
        """The estimator on one row layout (ragged with ``plan``, dense [B, Nmax] without); scores requires grad.
        ``noise`` [B, Nmax] Gumbel(0, 0.3) draw or None for the in-kernel Philox stream of ``seed``."""
        from ... import autograd
        k = self._scheme_k()
        if self.nb_samples != 1 or k is None:
            raise NotImplementedError("backward needs nb_samples == 1 and the stock threshold top-k solver")
        if self.kind == "imle":
            t = self.target_distribution
            return autograd.imle_topk(scores, int(k), plan, noise, seed, float(t.alpha), float(t.beta),
                                      self.noise_temperature, self.target_noise_temperature)
        from .target_aimle import AdaptiveTargetDistribution
        if not isinstance(self.target_distribution, AdaptiveTargetDistribution):
            raise NotImplementedError("AIMLE backward is implemented for the adaptive target (masking.py:258-260)")
        return autograd.aimle_topk(scores, int(k), plan, noise, seed, self.target_distribution,
                                   self.noise_temperature, self.target_noise_temperature)

    def __call__(self, theta: Tensor, *args, noise: Optional[Tensor] = None):
        if theta.dim() != 3:
            raise ValueError(f"expected theta [B, Nmax, 1], got {tuple(theta.shape)}")
        B, nmax, ens = theta.shape
        S = self.nb_samples
        if noise is None and self.noise_distribution is not None and (self.noise_temperature != 0.0 or ops._rec(theta)):
            noise = self.noise_distribution.sample(torch.Size([B, S, nmax, ens])).to(theta.device)
        k = self._scheme_k()
        if ops._rec(theta):
            if ens != 1:
                raise NotImplementedError("ensemble dimension must be 1")
            nz = None if noise is None else noise.reshape(B, nmax).contiguous().float()
            z = self.differentiable(theta.reshape(B, nmax), None, nz).view(B, nmax, 1)
            aux = None
        elif S == 1 and ens == 1 and k is not None:
            dense = theta.detach().reshape(B, nmax).contiguous()
            z = ops.topk_threshold(dense, k, noise=None if (noise is None or self.noise_temperature == 0.0)
                                   else noise.reshape(B, nmax).contiguous().float(),
                                   noise_scale=self.noise_temperature if noise is not None else 0.0)
            z = z.view(B, nmax, 1)
            aux = None
        else:   # generic solver: perturb with torch elementwise ops, then call it
            pert = theta.detach()[:, None, ...].expand(B, S, nmax, ens)
            if noise is not None:
                pert = pert + noise * self.noise_temperature
            z, aux = self.function(pert.reshape(B * S, nmax, ens))
        if self.kind == "imle":
            return z.view(B, S, nmax, ens).permute(1, 0, 2, 3), aux        # wrapper.py:118
        return z                                                            # aimle.py:138


def _decorate(kind: str, function, target_distribution, noise_distribution, nb_samples, temperature,
              target_noise_temperature):
    def build(fn):
        return PerturbAndMAP(fn, noise_distribution, nb_samples, temperature, kind, target_distribution,
                             target_noise_temperature)
    return build if function is None else build(function)


def imle(function=None, target_distribution=None, noise_distribution=None, nb_samples: int = 1,
         input_noise_temperature: float = 1.0, target_noise_temperature: float = 1.0):
    """Decorator with the reference's keyword surface (wrapper.py:16-23)."""
    return _decorate("imle", function, target_distribution, noise_distribution, nb_samples,
                     input_noise_temperature, target_noise_temperature)


def aimle(function=None, target_distribution=None, noise_distribution=None, nb_samples: int = 1,
          nb_marginal_samples: int = 1, theta_noise_temperature: float = 1.0, target_noise_temperature: float = 1.0,
          symmetric_perturbation: bool = False, _is_minimization: bool = False):
    """Decorator with the reference's keyword surface (aimle.py:16-26)."""
    if nb_marginal_samples != 1:
        raise NotImplementedError("nb_marginal_samples > 1 is never used by ISubGVQA (masking.py:258-281)")
    return _decorate("aimle", function, target_distribution, noise_distribution, nb_samples,
                     theta_noise_temperature, target_noise_temperature)
