"""Node gate + discrete top-k node mask.

Reference behaviour: MaskingModel, ISubGVQA/models/masking.py:23-199, and the sampler factories
:214-283.  Per node: x' = gelu(node_nn(x)); gate = gelu(<x'_n, ques_nn(u)[batch]_n>/sqrt(C)); the gates
are padded per graph to Nmax with 0.0 (pads compete), a sampler picks a k-hot mask per
row, and the real slots are gathered back to [N,1].

Kernels: isg_node_gate, then isg_topk_gumbel / isg_topk_threshold in their ragged row layout, which
fuses to_dense_batch (masking.py:162), the sampler and the `[mask]` un-pad (:170-176).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from .. import ops
from ..sampling.methods.deterministic_scheme import IMLEScheme
from ..sampling.methods.gumbel_scheme import GumbelSampler
from ..sampling.methods.noise import GumbelDistribution
from .layers import TopKPoolingParams

_IMLE_NOISE_SCALE = 0.3    # masking.py:215,249


class MaskingModel(torch.nn.Module):
    def __init__(self, dim_nodes, dim_questions, masking_threshold=0.3, use_topk=False, sample_k=None,
                 sampler_type=None, nb_samples=1, alpha=1.0, beta=10.0, tau=1.0):
        super().__init__()
        self.use_topk = use_topk
        self.sample_k = sample_k
        self.sampler_type = sampler_type
        self.masking_threshold = int(masking_threshold) if masking_threshold > 1 else masking_threshold
        self.dim_nodes, self.dim_questions = dim_nodes, dim_questions
        self.nb_samples, self.tau = nb_samples, tau
        self.gate_dropout = 0.2                                                          # masking.py:159,196

        self.gate_nn = torch.nn.Sequential(torch.nn.Linear(dim_questions, dim_questions), torch.nn.GELU(),
                                           torch.nn.Linear(dim_questions, 1))          # unused in forward (:139)
        self.node_nn = torch.nn.Sequential(torch.nn.Linear(dim_nodes, dim_questions), torch.nn.GELU())
        self.ques_nn = torch.nn.Sequential(torch.nn.Linear(dim_questions, dim_questions), torch.nn.GELU())
        if use_topk:
            self.gate_top = TopKPoolingParams(dim_questions)

        if sampler_type == "imle":
            self.sampler_train, self.sampler_val = get_imle_samplers(
                sample_k=sample_k, device="cuda", nb_samples=nb_samples, alpha=alpha, beta=beta, tau=tau)
        elif sampler_type == "aimle":
            self.sampler_train, self.sampler_val = get_aimle_samplers(
                sample_k=sample_k, device="cuda", nb_samples=nb_samples, alpha=alpha, tau=tau)
        elif sampler_type == "gumbel":
            self.sampler = GumbelSampler(k=sample_k, policy="edge_candid", train_ensemble=1, val_ensemble=1)
        elif sampler_type == "simple":
            from ..sampling.methods.simple_scheme import EdgeSIMPLEBatched
            self.sampler = EdgeSIMPLEBatched(k=sample_k, device="cuda", policy="edge_candid")    # masking.py:110-119

    def reset_parameters(self):
        for seq in (self.gate_nn, self.node_nn, self.ques_nn):
            for m in seq:
                if hasattr(m, "reset_parameters"):
                    m.reset_parameters()

    def planes_ready(self, u: Tensor) -> bool:
        """Will gate_scores() run on the layer input's planes (isg_node_gate_planes)?  Then no fp32 copy of the input is needed."""
        return (u is not None and u.dim() == 2 and u.size(1) == self.dim_questions == 128 and self.dim_nodes == 128
                and u.dtype == torch.float32 and ops.node_gate_planes_supported(self.node_nn, u))

    def gate_scores(self, x: Tensor, u: Tensor, batch: Tensor, u_is_per_graph: bool = False, plan=None,
                    x_planes=None) -> Tensor:
        """masking.py:137,151-155 -> [N,1].  ``u_is_per_graph``: u is [B,C] and the caller would have passed
        u[batch]; the reference then indexes ques_nn(u[batch]) with batch AGAIN (quirk Q3), which equals
        ques_nn(u)[batch[batch]] row for row -- computed here without the N-row GEMM."""
        q = ops.mlp(self.ques_nn, u)
        if x_planes is not None and self.planes_ready(u):
            # node_nn + the reduction against q as one launch on the planes the convolution reads anyway
            return ops.node_gate_planes(x_planes, self.node_nn, q.contiguous(), batch, double_index=u_is_per_graph)
        xn = ops.mlp(self.node_nn, x)
        return ops.node_gate(xn.contiguous(), q.contiguous(), batch, double_index=u_is_per_graph,
                             plan=plan)

    def forward(self, x, u, batch, edge_index, size=None, use_all_instrs=True, plan: Optional[ops.GraphPlan] = None,
                noise: Optional[Tensor] = None, seed: Optional[int] = None, u_is_per_graph: bool = False, x_planes=None):
        if use_all_instrs:
            raise NotImplementedError("use_all_instrs=True (masking.py:141-149) is off by default and outside this path")
        if x is not None:
            x = x.unsqueeze(-1) if x.dim() == 1 else x
        if plan is None:
            plan = ops.GraphPlan.build(batch, None, num_graphs=size)
        gate = self.gate_scores(x, u, batch, u_is_per_graph, plan, x_planes=x_planes)
        if not self.use_topk:                                               # masking.py:195-198
            gate = F.dropout(gate, p=self.gate_dropout, training=self.training)
            return (torch.sigmoid(gate) > 0.5).to(dtype=gate.dtype)
        gate = F.dropout(gate, p=self.gate_dropout, training=self.training)  # :159
        B, nmax = plan.B, plan.nmax
        if self.sampler_type == "gumbel":
            if noise is None and seed is None:                               # torch generator, like the reference
                from ..sampling.methods.noise import gumbel_from_uniform
                noise = gumbel_from_uniform(torch.rand(B, nmax, device=gate.device))
            return ops.topk_gumbel(gate, int(self.sample_k), float(self.sampler.tau), plan=plan, noise=noise,
                                   seed=0 if seed is None else seed)
        if self.sampler_type == "simple":          # masking.py:175-176 (same call as the Gumbel sampler)
            n = 1 << max(nmax - 1, 0).bit_length()
            if noise is None and seed is None:
                noise = torch.rand(B, n, device=gate.device)
            return ops.simple_topk(gate, int(self.sample_k), plan=plan, uniform=noise, seed=0 if seed is None else seed)
        if self.sampler_type in ("imle", "aimle"):
            sampler = self.sampler_train if self.training else self.sampler_val
            temp = sampler.noise_temperature
            if self.training and ops._rec(gate):       # estimator with the second MAP solve in its backward
                if noise is None and seed is None:
                    noise = sampler.noise_distribution.sample(torch.Size([B, 1, nmax, 1])).to(gate.device)
                nz = None if noise is None else noise.reshape(B, nmax).contiguous().float()
                return sampler.differentiable(gate, plan, nz, 0 if seed is None else seed)
            if temp == 0.0:
                return ops.topk_threshold(gate, int(self.sample_k), plan=plan)
            if noise is None and seed is None:
                noise = sampler.noise_distribution.sample(torch.Size([B, 1, nmax, 1])).to(gate.device)
            return ops.topk_threshold(gate, int(self.sample_k), plan=plan, noise=noise, noise_scale=temp,
                                      seed=0 if seed is None else seed)
        raise NotImplementedError(f"sampler_type={self.sampler_type!r}")


def _scheme_fn(scheduler: IMLEScheme):
    def solve(logits: Tensor):
        return scheduler.torch_sample_scheme(logits)
    solve._isg_threshold_k = scheduler.k      # lets the wrapper fuse perturbation + solve into one launch
    return solve


def get_imle_samplers(sample_k, beta=10, alpha=1.0, tau=1.0, noise_scale=_IMLE_NOISE_SCALE, nb_samples=1, device=None):
    """(train, eval) I-MLE samplers with the reference's settings (masking.py:214-245): the eval sampler
    multiplies its noise by 0 unless nb_samples > 1."""
    from ..sampling.methods.perturb_and_map import imle as _imle
    scheduler = IMLEScheme("edge_candid", sample_k, 1, 1)
    noise = GumbelDistribution(0.0, noise_scale, device)
    train = _imle(_scheme_fn(scheduler), target_distribution=("imle", alpha, beta), noise_distribution=noise,
                  nb_samples=nb_samples, input_noise_temperature=tau, target_noise_temperature=tau)
    val = _imle(_scheme_fn(scheduler), target_distribution=None, noise_distribution=noise, nb_samples=nb_samples,
                input_noise_temperature=tau if nb_samples > 1 else 0.0, target_noise_temperature=tau)
    return train, val


def get_aimle_samplers(sample_k, alpha=1.0, tau=1.0, noise_scale=_IMLE_NOISE_SCALE, nb_samples=1, device=None):
    """(train, eval) AIMLE samplers (masking.py:248-283): the eval sampler keeps theta temperature tau,
    i.e. it is stochastic at inference."""
    from ..sampling.methods.perturb_and_map import aimle as _aimle
    scheduler = IMLEScheme("edge_candid", sample_k, 1, 1)
    noise = GumbelDistribution(0.0, noise_scale, device)
    train = _aimle(_scheme_fn(scheduler), target_distribution=("aimle", alpha, 0.0), noise_distribution=noise,
                   nb_samples=nb_samples, theta_noise_temperature=tau, target_noise_temperature=tau,
                   symmetric_perturbation=True)
    val = _aimle(_scheme_fn(scheduler), target_distribution=None, noise_distribution=noise, nb_samples=nb_samples,
                 theta_noise_temperature=1.0 if nb_samples > 1 else tau, target_noise_temperature=tau,
                 symmetric_perturbation=True)
    return train, val
