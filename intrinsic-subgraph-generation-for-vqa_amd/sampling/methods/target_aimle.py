"""Target distributions of AIMLE.

Reference behaviour: ISubGVQA/sampling/methods/target_aimle.py:30-162.  AdaptiveTargetDistribution is
autograd.AdaptiveTarget: same update rule, but beta / grad_norm live on the device so a backward never synchronises.
"""
from __future__ import annotations

from typing import Optional

from torch import Tensor

from ...autograd import AdaptiveTarget as AdaptiveTargetDistribution  # noqa: F401


class TargetDistribution:
    """Fixed (alpha, beta) target (target_aimle.py:30-85)."""

    def __init__(self, alpha: float = 1.0, beta: float = 1.0, do_gradient_scaling: bool = False, eps: float = 1e-7):
        self.alpha, self.beta, self.do_gradient_scaling, self.eps = alpha, beta, do_gradient_scaling, eps

    def params(self, theta: Tensor, dy: Optional[Tensor], alpha: Optional[float] = None, beta: Optional[float] = None,
               _is_minimization: bool = False) -> Tensor:
        a = self.alpha if alpha is None else alpha
        b = self.beta if beta is None else beta
        d = 0.0 if dy is None else dy
        return a * theta + b * d if _is_minimization else a * theta - b * d

    def process(self, theta: Tensor, dy: Tensor, gradient_3d: Tensor) -> Tensor:
        return gradient_3d / max(self.beta, self.eps) if self.do_gradient_scaling else gradient_3d
