"""Target distribution of I-MLE: theta' = alpha * theta - beta * dy.

Reference behaviour: TargetDistribution, ISubGVQA/sampling/methods/target.py:22-44.  The arithmetic runs inside
autograd._ImleTopK.backward; this class carries (alpha, beta) under the reference's name.
"""
from __future__ import annotations

from torch import Tensor


class BaseTargetDistribution:
    def params(self, theta: Tensor, dy: Tensor) -> Tensor:
        raise NotImplementedError


class TargetDistribution(BaseTargetDistribution):
    def __init__(self, alpha: float = 1.0, beta: float = 1.0):
        self.alpha = alpha
        self.beta = beta

    def params(self, theta: Tensor, dy: Tensor) -> Tensor:
        return self.alpha * theta - self.beta * dy
