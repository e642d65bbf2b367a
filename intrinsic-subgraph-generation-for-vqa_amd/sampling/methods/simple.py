"""The exactly-k constraint circuit behind the SIMPLE sampler.

Reference behaviour: Layer, ISubGVQA/sampling/methods/simple.py:120-252, over the circuit of
create_simple_constraint.py:34-73 (pickled to ./simple_configs/{n}C{k}.pkl there; nothing is pickled here).
Forward = isg_simple_topk (csrc/isg_simple.hip): marginals and the Gumbel top-k sample in one launch per batch.
`log_marginals` below is the same function written with differentiable torch ops ON THE DEVICE; it is only evaluated in
the backward pass (autograd._Recomputed), where the reference also relies on autograd through its levelwise tensors.
The circuit is a balanced binary tree; every block of a level is alike, so its structure is three small (level, count)
tables -- see the kernel's header for the two dummy-padding accidents that are part of the function.
"""
from __future__ import annotations

import math
from functools import lru_cache
import torch
from torch import Tensor

LARGE_NUMBER = 1.0e10      # simple_scheme.py:16
_DUMMY = -1000.0           # simple.py:219


@lru_cache(maxsize=None)
def circuit_tables(n: int, k: int):
    """(levels, cap[l], reach[l][j], max_elements, max_parents) of the exactly-k circuit over n = 2^levels variables."""
    L = int(math.log2(n))
    if 2 ** L != n or not 0 < k <= n:
        raise ValueError(f"need n a power of two and 0 < k <= n, got n={n}, k={k}")
    cap = [min(k, 2 ** l) for l in range(L + 1)]
    elems = lambda l, j: [jj for jj in range(j + 1) if jj <= cap[l - 1] and j - jj <= cap[l - 1]]
    reach = [[False] * (k + 1) for _ in range(L + 1)]
    reach[L][k] = True
    for l in range(L - 1, -1, -1):
        for j in range(cap[l] + 1):
            reach[l][j] = any(reach[l + 1][jp] and 0 <= jp - j <= cap[l] for jp in range(cap[l + 1] + 1))
    max_el = max([len(elems(l, j)) for l in range(1, L + 1) for j in range(cap[l] + 1) if reach[l][j]] or [0])
    n_par = lambda l, j: sum(1 for jp in range(cap[l + 1] + 1) if reach[l + 1][jp] and 0 <= jp - j <= cap[l])
    max_par = max([n_par(l, j) for l in range(L) for j in range(cap[l] + 1) if reach[l][j]] or [0])
    return L, cap, reach, max_el, max_par


def _log1mexp(x: Tensor) -> Tensor:
    x = -x.abs()
    return torch.where(x > -0.6931471805599453094, torch.log(-torch.expm1(x)), torch.log1p(-torch.exp(x)))


def log_marginals(log_probs: Tensor, k: int) -> Tensor:
    """Layer.log_pr (simple.py:203-236): log_probs [R, n] -> log-marginals [R, n]; differentiable."""
    R, n = log_probs.shape
    L, cap, reach, max_el, max_par = circuit_tables(n, k)
    new = lambda *shape, v=float("-inf"): torch.full(shape, v, dtype=log_probs.dtype, device=log_probs.device)
    put = lambda t, j, v: torch.cat([t[:, :, :j], v.unsqueeze(-1), t[:, :, j + 1:]], dim=-1)
    W = [torch.cat([torch.stack((_log1mexp(-log_probs.detach()), log_probs), dim=-1), new(R, n, max(k - 1, 0))], dim=-1)]
    thetas = []
    for l in range(1, L + 1):
        left, right, blocks = W[l - 1][:, 0::2, :], W[l - 1][:, 1::2, :], n >> l
        w_l, th_l = new(R, blocks, k + 1), {}
        for j in range(cap[l] + 1):
            if not reach[l][j]:
                continue
            jjs = [jj for jj in range(j + 1) if jj <= cap[l - 1] and j - jj <= cap[l - 1]]
            terms = [left[:, :, jj] + right[:, :, j - jj] for jj in jjs]
            tot = torch.logsumexp(torch.stack(terms + [new(R, blocks, v=2 * _DUMMY)] * (max_el - len(jjs)), dim=-1), dim=-1)
            w_l = put(w_l, j, tot)
            th_l[j] = (jjs, [t - tot for t in terms])
        W.append(w_l)
        thetas.append(th_l)
    M = new(R, 1, k + 1)
    M = put(M, k, torch.zeros(R, 1, dtype=log_probs.dtype, device=log_probs.device))
    for l in range(L - 1, -1, -1):
        blocks = n >> l
        m_l = new(R, blocks, k + 1)
        for j in range(cap[l] + 1):
            if not reach[l][j]:
                continue
            contrib = []
            for jp in range(cap[l + 1] + 1):
                if reach[l + 1][jp] and 0 <= jp - j <= cap[l]:
                    jjs, th = thetas[l][jp]
                    up = M[:, :, jp]
                    contrib.append(torch.stack((th[jjs.index(j)] + up, th[jjs.index(jp - j)] + up), dim=-1).reshape(R, blocks))
            stack = torch.stack(contrib + [new(R, blocks, v=_DUMMY)] * (max_par - len(contrib)), dim=-1)
            m_l = put(m_l, j, torch.logsumexp(stack, dim=-1))
        M = m_l
    return M[:, :, 1]


class Layer:
    """Reference-shaped handle (simple.py:120): ``Layer(n, k, device)``; ``log_pr`` returns [n, R] like the reference."""

    def __init__(self, n, k, device=None, root="./simple_configs"):
        self.n, self.k, self.device = n, k, device
        circuit_tables(n, k)

    def log_pr(self, log_probs: Tensor) -> Tensor:
        return log_marginals(log_probs, self.k).permute(1, 0)
