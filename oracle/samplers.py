"""Discrete top-k node-mask samplers, restated on torch CPU ops (forward, and the training-mode backward rules).

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
        res = khot_hard - khot.detach() + khot                   # :88 straight-through
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


# ---------------------------------------------------------------------------
# Training mode: the custom backward rules of the perturb-and-MAP wrappers
# ---------------------------------------------------------------------------
class ImleTrain(torch.autograd.Function):
    """imle wrapper, nb_samples=1, TargetDistribution(alpha, beta) (wrapper.py:75-172, target.py:40-44).

    forward: z = MAP(theta + noise*tau_in) -> [1, B, Nmax, 1];
    backward: z' = MAP(alpha*theta - beta*dy + noise*tau_target); grad = z - z'.
    """

    @staticmethod
    def forward(ctx, theta, noise, k, alpha, beta, tau_in, tau_target):
        B, Nmax, ens = theta.shape
        pert = theta[:, None, ...] + noise * tau_in                                 # :93-100
        z = threshold_topk(pert.view(B, Nmax, ens), k).view(B, 1, Nmax, ens)        # :103-110
        ctx.save_for_backward(theta, noise, z)
        ctx.cfg = (k, alpha, beta, tau_target)
        return z.permute(1, 0, 2, 3)                                                # :118

    @staticmethod
    def backward(ctx, dy):
        theta, noise, z = ctx.saved_tensors
        k, alpha, beta, tau_target = ctx.cfg
        B, Nmax, ens = theta.shape
        dy = dy.permute(1, 0, 2, 3)                                                 # :136
        target = alpha * theta[:, None, ...] - beta * dy                            # :146-147, target.py:43
        pert = target + noise * tau_target                                          # :153-156
        z_t = threshold_topk(pert.reshape(B, Nmax, ens), k).view(B, 1, Nmax, ens)   # :164
        return (z - z_t).mean(dim=1), None, None, None, None, None, None           # :170-172


class AimleTargetState:
    """AdaptiveTargetDistribution (target_aimle.py:88-162): beta adapts so that ~target_norm entries flip."""

    def __init__(self, alpha: float = 1.0, beta: float = 0.0, grad_norm: float = 1.0, step: float = 1e-4,
                 momentum: float = 0.0, decay: float = 0.9, target_norm: float = 1.0):
        self.alpha, self.beta, self.grad_norm = alpha, beta, grad_norm
        self.step, self.momentum, self.decay, self.target_norm = step, momentum, decay, target_norm
        self.previous_update = 0.0

    def magnitude(self, theta: Tensor, dy: Tensor):
        norm_dy = torch.linalg.norm(dy).item()                                      # :113
        return 0.0 if norm_dy <= 0.0 else self.beta * (torch.linalg.norm(theta) / norm_dy)   # :114-116

    def params(self, theta: Tensor, dy: Tensor) -> Tensor:
        return self.alpha * theta - self.magnitude(theta, dy) * dy                  # :124-129

    def process(self, theta: Tensor, dy: Tensor, grad: Tensor) -> Tensor:
        pm = self.magnitude(theta, dy)                                              # :134 (with the OLD beta)
        nnz = torch.count_nonzero(grad).float()
        self.grad_norm = self.decay * self.grad_norm + (1.0 - self.decay) * (nnz / (grad.shape[0] * grad.shape[1]))
        upd = (1.0 if float(self.grad_norm) < self.target_norm else -1.0) * self.step   # :149-151
        upd = self.momentum * self.previous_update + upd
        self.beta = max(self.beta + upd, 0.0)                                       # :157
        self.previous_update = upd
        return grad / (pm if pm > 0.0 else 1.0)                                     # :161


class AimleTrain(torch.autograd.Function):
    """aimle wrapper, nb_samples=1, symmetric perturbation (aimle.py:83-243).

    forward: z = MAP(theta + noise*tau_theta) [B, Nmax, 1];
    backward: z_R = MAP(theta'_R + eps), z_L = MAP(theta'_L + eps) with theta'_{R,L} = alpha*theta -/+ lambda*dy,
    grad = (z_L - z_R)/2 / lambda, and the target state is updated.
    """

    @staticmethod
    def forward(ctx, theta, noise, k, state, tau_theta, tau_target):
        B, Nmax, ens = theta.shape
        pert = theta.view(B, 1, Nmax, ens) + noise * tau_theta                      # :109-117
        z = threshold_topk(pert.view(B, Nmax, ens), k)
        ctx.save_for_backward(theta, noise)
        ctx.cfg = (k, state, tau_target)
        return z

    @staticmethod
    def backward(ctx, dy):
        theta, noise = ctx.saved_tensors
        k, state, tau_target = ctx.cfg
        B, Nmax, ens = theta.shape
        t_r = state.params(theta, dy)                                               # :173-176
        t_l = state.params(theta, -dy)                                              # :178-182
        eps = (noise * tau_target).view(B, Nmax, ens)                               # :189
        z_r = threshold_topk(t_r + eps, k)                                          # :203
        z_l = threshold_topk(t_l + eps, k)                                          # :206
        g = ((z_l - z_r) / 2.0).view(B, 1, Nmax, ens)                               # :231-234
        g = state.process(theta, dy, g)                                             # :237
        return g.mean(dim=1), None, None, None, None, None                          # :240-242
