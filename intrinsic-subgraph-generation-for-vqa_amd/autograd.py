"""Training of the hot path: autograd wiring around the HIP kernels.

`ops.*` dispatch here when autograd is recording and an input requires grad; inside a Function's forward autograd is
off, so the same `ops.*` call runs the plain forward kernel.  Three kinds of backward:

  * hand-written HIP kernels for the operators the reference differentiates by hand or that dominate the step:
      GatV2MP            isg_gatv2_mp_bwd            (PyG propagate autograd behind mgat_v2_conv.py:215-279)
      NodeToEdgeMask     isg_node_to_edge_mask_bwd   (NodeMaskToEdgeMask.backward, sampling/node_edge_masks.py:13-19)
      GumbelTopK         isg_topk_gumbel_bwd         (straight-through, gumbel_scheme.py:83-90)
      ImleTopK/AimleTopK isg_topk_threshold again    (second MAP solve, wrapper.py:124-172; aimle.py:141-243;
                                                      adaptive beta, target_aimle.py:88-162 -- state kept ON THE DEVICE)
  * Linear: forward and dX on the bf16x6 matrix-core kernel, dW (a reduction over the rows) as a split-M GEMM on the
    fp32 matrix-core instruction (csrc/isg_wgrad.hip);
  * the per-graph operators around the message passing -- layer tail (instruction attention + GraphNorm + residual),
    pooling, instruction gate, node gate: per-graph HIP backward kernels (csrc/isg_tail_bwd.hip);
  * what only the scene-graph encoder / stand-alone utilities use (GraphNorm alone, scatter attention alone,
    scatter_mean, the SIMPLE marginals): forward = fused kernel, backward re-evaluates a torch-op restatement ON THE
    DEVICE under autograd (`_Recomputed`).  These are plain autograd in the reference too.

Nothing here runs on the CPU and nothing imports `oracle/`.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import ops


# ------------------------------------------------------------------------------------------------
# Generic: fused forward, recomputed torch backward
# ------------------------------------------------------------------------------------------------
class _Recomputed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fused, restate, consts, *tensors):
        ctx.restate, ctx.consts = restate, consts
        ctx.save_for_backward(*tensors)
        out = fused(*tensors, *consts)
        return out

    @staticmethod
    def backward(ctx, *gouts):
        tensors = ctx.saved_tensors
        need = ctx.needs_input_grad[3:]
        with torch.enable_grad():
            ins = [None if t is None else (t.detach().requires_grad_(True) if n else t.detach())
                   for t, n in zip(tensors, need)]
            outs = ctx.restate(*ins, *ctx.consts)
            outs = outs if isinstance(outs, tuple) else (outs,)
            pairs = [(o, g) for o, g in zip(outs, gouts) if g is not None and o.requires_grad]
            wanted = [i for i, n in zip(ins, need) if n]
            grads = torch.autograd.grad([o for o, _ in pairs], wanted, [g for _, g in pairs], allow_unused=True)
        it = iter(grads)
        return (None, None, None) + tuple(next(it) if n else None for n in need)


def _seg_sum(v: Tensor, batch: Tensor, B: int) -> Tensor:
    return torch.zeros((B,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device).index_add_(0, batch, v)


def _seg_softmax(logits: Tensor, batch: Tensor, B: int, eps: float = 0.0) -> Tensor:
    m = torch.full((B,), -math.inf, dtype=logits.dtype, device=logits.device)
    m = m.scatter_reduce(0, batch, logits.detach(), "amax", include_self=True)
    e = (logits - m[batch]).exp()
    return e / (_seg_sum(e, batch, B)[batch] + eps)


def _instr_gate_t(x, instr, batch):
    return F.gelu(x * instr[batch])


class _InstrGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, instr, batch, plan):
        ctx.save_for_backward(x, instr)
        ctx.plan = plan
        return ops.instr_gate(x, instr, batch)

    @staticmethod
    def backward(ctx, g):
        x, instr = ctx.saved_tensors
        d_x, d_instr = ops.instr_gate_backward(x, instr, ctx.plan, g.contiguous())
        return d_x, d_instr, None, None


def instr_gate(x, instr, batch, plan=None):
    if plan is not None and instr.size(0) == plan.B:
        return _InstrGate.apply(x, instr, batch, plan)
    return _Recomputed.apply(ops.instr_gate, _instr_gate_t, (batch,), x, instr)


def _node_gate_t(xn, q, batch, double_index):
    idx = batch[batch] if double_index else batch
    return F.gelu((xn * q[idx]).sum(-1, keepdim=True) / math.sqrt(xn.size(1)))


class _NodeGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xn, q, batch, double_index, plan):
        ctx.save_for_backward(xn, q, batch)
        ctx.cfg = (double_index, plan)
        return ops.node_gate(xn, q, batch, double_index)

    @staticmethod
    def backward(ctx, g):
        xn, q, batch = ctx.saved_tensors
        double_index, plan = ctx.cfg
        d_xn, d_q = ops.node_gate_backward(xn, q, batch, double_index, plan, g.contiguous())
        return d_xn, d_q, None, None, None


def node_gate(xn, q, batch, double_index, plan=None):
    if plan is not None:
        return _NodeGate.apply(xn, q, batch, double_index, plan)
    return _Recomputed.apply(ops.node_gate, _node_gate_t, (batch, double_index), xn, q)


def _graph_norm_t(v, weight, bias, mean_scale, batch, B, eps, fp64):
    dt = v.dtype
    if fp64:
        v, weight, bias, mean_scale = v.double(), weight.double(), bias.double(), mean_scale.double()
    cnt = torch.bincount(batch, minlength=B).clamp(min=1).to(v.dtype).unsqueeze(1)
    mean = _seg_sum(v, batch, B) / cnt
    out = v - mean[batch] * mean_scale
    var = _seg_sum(out * out, batch, B) / cnt
    return (weight * out / (var + eps).sqrt()[batch] + bias).to(dt)


def _layer_tail_t(ins, c, h, weight, bias, mean_scale, node_mask, batch, B, eps):
    att = _seg_softmax((ins[batch] * c).sum(-1) / math.sqrt(c.size(1)), batch, B)
    y = _graph_norm_t(att.unsqueeze(1) * c, weight, bias, mean_scale, batch, B, eps, False) + h
    return y if node_mask is None else node_mask.view(-1, 1) * y


def _layer_tail_f(ins, c, h, weight, bias, mean_scale, node_mask, plan, eps):
    return ops.mgat_layer_tail(ins, c, h, plan, weight, bias, mean_scale, eps, node_mask=node_mask)


class _LayerTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ins, c, h, weight, bias, mean_scale, node_mask, plan, eps):
        ctx.save_for_backward(ins, c, h, weight, bias, mean_scale, node_mask)
        ctx.cfg = (plan, eps)
        return ops.mgat_layer_tail(ins, c, h, plan, weight, bias, mean_scale, eps, node_mask=node_mask)

    @staticmethod
    def backward(ctx, g):
        ins, c, h, weight, bias, mean_scale, node_mask = ctx.saved_tensors
        plan, eps = ctx.cfg
        want_mask = node_mask is not None and ctx.needs_input_grad[6]
        d_ins, d_c, d_h, d_w, d_b, d_ms, d_m = ops.layer_tail_backward(
            ins, c, h, plan, weight, bias, mean_scale, eps, node_mask, g.contiguous(), want_mask)
        return d_ins, d_c, d_h, d_w, d_b, d_ms, (d_m.view_as(node_mask) if want_mask else None), None, None


def mgat_layer_tail(ins, c, h, plan, weight, bias, mean_scale, eps, node_mask):
    return _LayerTail.apply(ins, c, h, weight, bias, mean_scale, node_mask, plan, eps)


def _pool_t(xn, q, node_mask, batch, B):
    x = xn if node_mask is None else xn * node_mask.view(-1, 1)
    gate = _seg_softmax((x * q[batch]).sum(-1) / math.sqrt(x.size(1)), batch, B, 1e-16).unsqueeze(1)
    return _seg_sum(gate * x, batch, B), gate


def _pool_f(xn, q, node_mask, plan):
    return ops.global_attn_pool(xn, q, plan, node_mask)


class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xn, q, node_mask, plan):
        ctx.save_for_backward(xn, q, node_mask)
        ctx.plan = plan
        ctx.set_materialize_grads(False)          # the gate output is usually unused: its gradient arrives as None
        return ops.global_attn_pool(xn, q, plan, node_mask)

    @staticmethod
    def backward(ctx, g_out, g_gate):
        xn, q, node_mask = ctx.saved_tensors
        if g_out is None:
            g_out = torch.zeros(ctx.plan.B, xn.size(1), dtype=xn.dtype, device=xn.device)
        want_mask = node_mask is not None and ctx.needs_input_grad[2]
        d_xn, d_q, d_m = ops.global_attn_pool_backward(xn, q, ctx.plan, node_mask, g_out.contiguous(),
                                                       None if g_gate is None else g_gate.contiguous(), want_mask)
        return d_xn, d_q, (d_m.view_as(node_mask) if want_mask else None), None


def global_attn_pool(xn, q, plan, node_mask):
    return _Pool.apply(xn, q, node_mask, plan)


def graph_norm(x, plan, weight, bias, mean_scale, eps, fp64):
    batch, B = plan.batch, plan.B
    return _Recomputed.apply(lambda v, w, b, m, p, e, f: ops.graph_norm(v, p, w, b, m, e, f),
                             lambda v, w, b, m, p, e, f: _graph_norm_t(v, w, b, m, batch, B, e, f),
                             (plan, eps, fp64), x, weight, bias, mean_scale)


def scatter_attention(query, key, plan, value):
    batch, B = plan.batch, plan.B

    def restate(qr, k, v, _p):
        att = _seg_softmax((qr[batch] * k).sum(-1) / math.sqrt(k.size(1)), batch, B)
        return att.unsqueeze(1) * v
    return _Recomputed.apply(lambda qr, k, v, p: ops.scatter_attention(qr, k, p, v), restate, (plan,), query, key, value)


def scatter_mean(msg, plan):
    dst = plan.edge_index[1]
    N = plan.N

    def restate(m, _p):
        cnt = torch.bincount(dst, minlength=N).clamp(min=1).to(m.dtype).unsqueeze(1)
        return _seg_sum(m, dst, N) / cnt
    return _Recomputed.apply(ops.scatter_mean, restate, (plan,), msg)


# ------------------------------------------------------------------------------------------------
# Dense projection
# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = act(x W^T + b): forward on isg_linear_bf16x6 (pre-activation kept when act = GELU), backward as fp32 GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, gelu):
        z = ops.linear(x, weight, bias, gelu=False, cache_planes=False)
        ctx.gelu = gelu
        ctx.save_for_backward(x, weight, z if gelu else None)
        ctx.has_bias = bias is not None
        return F.gelu(z) if gelu else z

    @staticmethod
    def backward(ctx, g):
        x, weight, z = ctx.saved_tensors
        g = g.contiguous()
        if ctx.gelu:
            g = torch.ops.aten.gelu_backward(g, z)
        dx = None
        if ctx.needs_input_grad[0]:     # dX = g W: the same matrix-core kernel on the transposed weight
            dx = (ops.linear(g, weight.detach().t().contiguous(), None, cache_planes=False)
                  if (g.size(1) & 3) == 0 else g @ weight)
        dw = None
        if ctx.needs_input_grad[1]:   # long-and-thin reductions on the split-M kernel; short ones are hipBLASLt's home turf
            dw = ops.linear_wgrad(g, x.contiguous()) if g.size(0) >= 16384 else g.t() @ x
        db = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def linear(x, weight, bias, gelu):
    return _Linear.apply(x, weight, bias, gelu)


# ------------------------------------------------------------------------------------------------
# Message passing and the node -> edge mask
# ------------------------------------------------------------------------------------------------
class _GatV2MP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_l, x_r, e_proj, att, bias, node_mask, edge_mask, plan, heads, slope, kernel):
        out, alpha = ops.gatv2_mp(x_l, x_r, e_proj, att, plan, heads, bias=bias, node_mask=node_mask,
                                  edge_mask=edge_mask, negative_slope=slope, kernel=kernel)
        ctx.save_for_backward(x_l, x_r, e_proj, att, alpha, node_mask, edge_mask)
        ctx.cfg = (plan, heads, slope, bias is not None)
        ctx.mark_non_differentiable(alpha)
        return out, alpha

    @staticmethod
    def backward(ctx, g_out, _g_alpha):
        x_l, x_r, e_proj, att, alpha, node_mask, edge_mask = ctx.saved_tensors
        plan, heads, slope, has_bias = ctx.cfg
        need = ctx.needs_input_grad
        want_mask = (node_mask is not None and need[5]) or (edge_mask is not None and need[6])
        d_xl, d_xr, d_e, d_att, d_bias, d_m = ops.gatv2_mp_backward(
            x_l, x_r, e_proj, att, alpha, g_out, plan, heads, node_mask=node_mask, edge_mask=edge_mask,
            negative_slope=slope, want_mask_grad=want_mask)
        d_node = d_edge = None
        if want_mask and node_mask is not None:       # the fused mask[src]*mask[dst] product keeps the reference's rule
            d_node = ops.node_to_edge_mask_backward(d_m, plan).view_as(node_mask)
        elif want_mask:
            d_edge = d_m.view_as(edge_mask)
        return (d_xl, d_xr, d_e, d_att.view_as(att), d_bias if has_bias else None, d_node, d_edge,
                None, None, None, None)


def gatv2_mp(x_l, x_r, e_proj, att, plan, heads, bias, node_mask, edge_mask, negative_slope, kernel):
    return _GatV2MP.apply(x_l, x_r, e_proj, att, bias, node_mask, edge_mask, plan, heads, negative_slope, kernel)


class _NodeToEdgeMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, edge_index, plan):
        ctx.plan = plan
        ctx.shape = mask.shape
        return ops.node_to_edge_mask(mask, edge_index)

    @staticmethod
    def backward(ctx, g):
        return ops.node_to_edge_mask_backward(g.contiguous(), ctx.plan).view(ctx.shape), None, None


def node_to_edge_mask(mask, edge_index, plan):
    return _NodeToEdgeMask.apply(mask, edge_index, plan)


# ------------------------------------------------------------------------------------------------
# Samplers
# ------------------------------------------------------------------------------------------------
class _GumbelTopK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, k, tau, plan, noise, seed):
        ctx.save_for_backward(scores, noise)
        ctx.cfg = (k, tau, plan, seed)
        return ops.topk_gumbel(scores, k, tau, plan=plan, noise=noise, seed=seed)

    @staticmethod
    def backward(ctx, g):
        scores, noise = ctx.saved_tensors
        k, tau, plan, seed = ctx.cfg
        return ops.topk_gumbel_backward(scores, g, k, tau, plan=plan, noise=noise, seed=seed), None, None, None, None, None


def topk_gumbel(scores, k, tau, plan, noise, seed):
    return _GumbelTopK.apply(scores, k, tau, plan, noise, seed)


def simple_topk(scores, k, plan, uniform, seed, return_marginals):
    """forward = isg_simple_topk; backward = autograd over the torch restatement of the circuit's marginals (the only
    differentiable part: out = (sample - marginals).detach() + marginals)."""
    from .sampling.methods.simple import LARGE_NUMBER, log_marginals
    if plan is not None:
        B, nmax, slots = plan.B, plan.nmax, plan.dense_slots()
    else:
        B, nmax, slots = scores.shape[0], scores.shape[1], None
    n = 1 << max(nmax - 1, 0).bit_length()
    kk = min(int(k), nmax)

    def fused(sc, *_c):
        return ops.simple_topk(sc, k, plan, uniform, seed, return_marginals)

    def restate(sc, *_c):
        if slots is not None:
            dense = torch.zeros(B * nmax, dtype=sc.dtype, device=sc.device).index_put((slots,), sc.reshape(-1)).view(B, nmax)
        else:
            dense = sc.reshape(B, nmax)
        flat = torch.cat([dense, torch.full((B, n - nmax), -LARGE_NUMBER, dtype=sc.dtype, device=sc.device)], dim=1)
        marg = log_marginals(flat, kk).exp()[:, :nmax]
        out = marg.reshape(-1)[slots].view(sc.shape) if slots is not None else marg.view(sc.shape)
        return (out, marg) if return_marginals else out

    return _Recomputed.apply(fused, restate, (), scores)


class _ImleTopK(torch.autograd.Function):
    """z = MAP(theta + tau_in * eps);  d theta = z - MAP(alpha * theta - beta * dy + tau_t * eps)."""

    @staticmethod
    def forward(ctx, scores, k, plan, noise, seed, alpha, beta, tau_in, tau_target):
        z = ops.topk_threshold(scores, k, plan=plan, noise=noise, noise_scale=tau_in, seed=seed)
        ctx.save_for_backward(scores, noise, z)
        ctx.cfg = (k, plan, seed, alpha, beta, tau_target)
        return z

    @staticmethod
    def backward(ctx, dy):
        scores, noise, z = ctx.saved_tensors
        k, plan, seed, alpha, beta, tau_target = ctx.cfg
        target = alpha * scores - beta * dy                                              # target.py:43
        z_t = ops.topk_threshold(target.contiguous(), k, plan=plan, noise=noise, noise_scale=tau_target, seed=seed)
        return (z - z_t,) + (None,) * 8                                                  # wrapper.py:170-172 (S = 1)


class AdaptiveTarget:
    """AdaptiveTargetDistribution (target_aimle.py:88-162) with beta / grad_norm as 0-dim DEVICE tensors, so the
    backward never synchronises (the reference calls .item() three times per backward)."""

    def __init__(self, initial_alpha: float = 1.0, initial_beta: float = 1.0, initial_grad_norm: float = 1.0,
                 beta_update_step: float = 0.0001, beta_update_momentum: float = 0.0, grad_norm_decay_rate: float = 0.9,
                 target_norm: float = 1.0):
        self.alpha = initial_alpha
        self._beta0, self._gn0 = float(initial_beta), float(initial_grad_norm)
        self.beta_t: Optional[Tensor] = None        # float64, like the reference's Python float
        self.grad_norm_t: Optional[Tensor] = None   # float32
        self.prev_update_t: Optional[Tensor] = None
        self.beta_update_step, self.beta_update_momentum = beta_update_step, beta_update_momentum
        self.grad_norm_decay_rate, self.target_norm = grad_norm_decay_rate, target_norm

    def _init(self, device):
        if self.beta_t is None or self.beta_t.device != device:
            self.beta_t = torch.tensor(self._beta0, dtype=torch.float64, device=device)
            self.grad_norm_t = torch.tensor(self._gn0, dtype=torch.float32, device=device)
            self.prev_update_t = torch.zeros((), dtype=torch.float64, device=device)

    @property
    def beta(self) -> float:
        return self._beta0 if self.beta_t is None else float(self.beta_t)

    @property
    def grad_norm(self) -> float:
        return self._gn0 if self.grad_norm_t is None else float(self.grad_norm_t)

    def magnitude(self, theta: Tensor, dy: Tensor) -> Tensor:
        self._init(theta.device)
        norm_dy = torch.linalg.norm(dy)
        pm = self.beta_t.float() * (torch.linalg.norm(theta) / norm_dy)                  # :114-116
        return torch.where(norm_dy > 0, pm, torch.zeros_like(pm))

    def update(self, grad_dense: Tensor, n_gradients: int) -> None:
        nnz = torch.count_nonzero(grad_dense).float()                                    # :137
        d = self.grad_norm_decay_rate
        self.grad_norm_t = d * self.grad_norm_t + (1.0 - d) * (nnz / n_gradients)        # :144-146
        step = torch.where(self.grad_norm_t < self.target_norm, self.beta_update_step, -self.beta_update_step)
        upd = self.beta_update_momentum * self.prev_update_t + step.double()             # :149-154
        self.beta_t = torch.clamp(self.beta_t + upd, min=0.0)                            # :157
        self.prev_update_t = upd


class _AimleTopK(torch.autograd.Function):
    """z = MAP(theta + tau * eps);  d theta = (MAP(theta'_L + eps') - MAP(theta'_R + eps')) / 2 / lambda with
    theta'_{R,L} = alpha * theta -/+ lambda * dy (symmetric perturbation).  The two target solves also return their
    selection over the padded rows: the adaptive rule counts the flipped slots INCLUDING the pads."""

    @staticmethod
    def forward(ctx, scores, k, plan, noise, seed, state, tau_theta, tau_target):
        z = ops.topk_threshold(scores, k, plan=plan, noise=noise, noise_scale=tau_theta, seed=seed)
        ctx.save_for_backward(scores, noise)
        ctx.cfg = (k, plan, seed, state, tau_target)
        return z

    @staticmethod
    def backward(ctx, dy):
        scores, noise = ctx.saved_tensors
        k, plan, seed, state, tau_target = ctx.cfg
        pm = state.magnitude(scores, dy)
        t_r = (state.alpha * scores - pm * dy).contiguous()                              # aimle.py:173-176
        t_l = (state.alpha * scores + pm * dy).contiguous()                              # :178-182 (params(theta, -dy))
        z_r, dense_r = ops.topk_threshold(t_r, k, plan=plan, noise=noise, noise_scale=tau_target, seed=seed,
                                          return_dense=True)
        z_l, dense_l = ops.topk_threshold(t_l, k, plan=plan, noise=noise, noise_scale=tau_target, seed=seed,
                                          return_dense=True)
        state.update((dense_l - dense_r) / 2.0, dense_l.size(0))                         # target_aimle.py:131-159
        g = (z_l - z_r) / 2.0 / torch.where(pm > 0, pm, torch.ones_like(pm))             # aimle.py:231-237, :161
        return (g,) + (None,) * 7


def imle_topk(scores, k, plan, noise, seed, alpha, beta, tau_in, tau_target):
    return _ImleTopK.apply(scores, k, plan, noise, seed, alpha, beta, tau_in, tau_target)


def aimle_topk(scores, k, plan, noise, seed, state, tau_theta, tau_target):
    return _AimleTopK.apply(scores, k, plan, noise, seed, state, tau_theta, tau_target)
