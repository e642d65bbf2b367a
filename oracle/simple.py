"""SIMPLE sampler (exact k-subset marginals through the reference's constraint circuit), restated on torch CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  SURVEY §8f row 4.

Follows ISubGVQA/sampling/methods/simple_scheme.py:44-162 (EdgeSIMPLEBatched.forward, policy 'edge_candid'),
simple.py:16-252 (Layer: levelwise log-probabilities up, marginals down, Gumbel top-k sample) and
create_simple_constraint.py:34-73 (the exactly-k circuit).  The circuit is a balanced binary tree over N = 2^ceil(log2 Nmax)
variables: node (level l, block i, count j) = OR_jj AND(node(l-1, 2i, jj), node(l-1, 2i+1, j-jj)); leaves are the
literals.  All blocks of a level are alike, so the reference's per-node index tensors reduce to small (level, count)
tables -- including two accidents of its padding that decide the result on ragged batches:
  * every node's element list is padded to `max_elements` with a dummy node whose log-weight is -1000, so an impossible
    node evaluates to about -2000 instead of -inf (simple.py:217-219,199-206);
  * every node's parent list is padded to `max_parents` with that dummy (log-marginal contribution -1000) (:151-160).
A zero score (the to_dense_batch pad, quirk Q1) has negative-literal weight log(1 - exp(-0)) = -inf, i.e. the slot is
FORCED into the subset; with more than k such pads the root is impossible and the marginals are finite garbage shaped by
the two paddings -- reproduced here, bit-compatible up to fp32 summation order.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import torch
from torch import Tensor

LARGE_NUMBER = 1.0e10          # simple_scheme.py:16
DUMMY = -1000.0                # simple.py:219  data[self.id] = -float(1000)


@dataclass
class Circuit:
    n: int                      # variables (power of two)
    k: int
    levels: int                 # log2(n)
    cap: List[int]              # cap[l] = min(k, 2^l): largest count a level-l node is built for
    reach: List[List[bool]]     # reach[l][j]: node (l, *, j) is reachable from the root (l = 0: j=1 positive, j=0 negative literal)
    n_elem: List[List[int]]     # elements of node (l, *, j), l >= 1
    n_par: List[List[int]]      # reachable parents of node (l, *, j)
    max_elements: int
    max_parents: int


def build_circuit(n: int, k: int) -> Circuit:
    L = int(math.log2(n))
    assert 2 ** L == n and 0 < k <= n
    cap = [min(k, 2 ** l) for l in range(L + 1)]
    n_elem = [[0] * (k + 1) for _ in range(L + 1)]
    for l in range(1, L + 1):
        for j in range(cap[l] + 1):
            n_elem[l][j] = sum(1 for jj in range(j + 1) if jj <= cap[l - 1] and j - jj <= cap[l - 1])
    reach = [[False] * (k + 1) for _ in range(L + 1)]
    reach[L][k] = True                                                 # create_and_save: alpha = dp[0][-1]
    for l in range(L - 1, -1, -1):
        for j in range(cap[l] + 1):
            reach[l][j] = any(reach[l + 1][jp] and 0 <= jp - j <= cap[l] for jp in range(cap[l + 1] + 1))
    n_par = [[0] * (k + 1) for _ in range(L + 1)]
    for l in range(L):
        for j in range(cap[l] + 1):
            n_par[l][j] = sum(1 for jp in range(cap[l + 1] + 1) if reach[l + 1][jp] and 0 <= jp - j <= cap[l])
    max_elements = max([n_elem[l][j] for l in range(1, L + 1) for j in range(cap[l] + 1) if reach[l][j]] or [0])
    max_parents = max([n_par[l][j] for l in range(L) for j in range(cap[l] + 1) if reach[l][j]] or [0])
    return Circuit(n, k, L, cap, reach, n_elem, n_par, max_elements, max_parents)


def log1mexp(x: Tensor) -> Tensor:
    """simple.py:45-57: log(1 - exp(-|x|))."""
    x = -x.abs()
    return torch.where(x > -0.6931471805599453094, torch.log(-torch.expm1(x)), torch.log1p(-torch.exp(x)))


def log_marginals(log_probs: Tensor, c: Circuit) -> Tensor:
    """Layer.log_pr (simple.py:203-236) for log_probs [R, n]; returns [R, n] log-marginals of the positive literals."""
    R, n = log_probs.shape
    k, L = c.k, c.levels
    NEG = float("-inf")
    # W[l]: [R, n / 2^l, k + 1]; entries of nodes that do not exist stay at -inf and are never read
    W = [torch.stack((log1mexp(-log_probs.detach()), log_probs), dim=-1)]             # :204-206 (neg. weights detached)
    if k > 1:
        W[0] = torch.cat([W[0], torch.full((R, n, k - 1), NEG)], dim=-1)
    terms_at = []
    for l in range(1, L + 1):
        left, right = W[l - 1][:, 0::2, :], W[l - 1][:, 1::2, :]
        blocks = n >> l
        w_l = torch.full((R, blocks, k + 1), NEG)
        terms_l = {}
        for j in range(c.cap[l] + 1):
            if not c.reach[l][j]:
                continue
            jjs = [jj for jj in range(j + 1) if jj <= c.cap[l - 1] and j - jj <= c.cap[l - 1]]
            terms = [left[:, :, jj] + right[:, :, j - jj] for jj in jjs]               # data[idx2primesub].sum(-2)  :26
            pads = c.max_elements - len(jjs)
            stack = torch.stack(terms + [torch.full((R, blocks), 2 * DUMMY)] * pads, dim=-1)
            tot = torch.logsumexp(stack, dim=-1)                                       # :27
            w_l = torch.cat([w_l[:, :, :j], tot.unsqueeze(-1), w_l[:, :, j + 1:]], dim=-1)
            terms_l[j] = (jjs, [t - tot for t in terms])                               # theta -= data  :28
        W.append(w_l)
        terms_at.append(terms_l)
    # marginals, top-down (levelwiseMars :33-42); M[L][.., k] = 0 (:230)
    M = [None] * (L + 1)
    M[L] = torch.full((R, 1, k + 1), NEG)
    M[L][:, :, k] = 0.0
    for l in range(L - 1, -1, -1):
        blocks = n >> l
        m_l = torch.full((R, blocks, k + 1), NEG)
        for j in range(c.cap[l] + 1):
            if not c.reach[l][j]:
                continue
            contrib = []
            for jp in range(c.cap[l + 1] + 1):
                if not (c.reach[l + 1][jp] and 0 <= jp - j <= c.cap[l]):
                    continue
                jjs, thetas = terms_at[l][jp]                    # terms_at[l] belongs to level l + 1
                up = M[l + 1][:, :, jp]                          # [R, blocks / 2]
                th_even = thetas[jjs.index(j)]                   # this child is the prime (left):  jj = j
                th_odd = thetas[jjs.index(jp - j)]               # this child is the sub (right):   jj = jp - j
                both = torch.stack((th_even + up, th_odd + up), dim=-1).reshape(R, blocks)
                contrib.append(both)
            pads = c.max_parents - len(contrib)
            stack = torch.stack(contrib + [torch.full((R, blocks), DUMMY)] * pads, dim=-1)
            m_l = torch.cat([m_l[:, :, :j], torch.logsumexp(stack, dim=-1).unsqueeze(-1), m_l[:, :, j + 1:]], dim=-1)
        M[l] = m_l
    return M[0][:, :, 1]


def simple_forward(scores: Tensor, k: int, uniform: Tensor):
    """EdgeSIMPLEBatched.forward, policy 'edge_candid', one sample, logits_activation None (simple_scheme.py:44-162).

    scores [B, Nmax, 1]; uniform [1, B, n] = the torch.rand draw of gumbel_keys (simple.py:99-104).
    Returns (new_mask [1, B, Nmax, 1], marginals [B, Nmax, 1])."""
    B, Nmax, ens = scores.shape
    assert ens == 1
    flat = scores.permute(0, 2, 1).reshape(B, Nmax)                                    # :82
    local_k = min(k, Nmax)                                                             # :84
    n = 2 ** math.ceil(math.log2(Nmax))                                                # :88
    c = build_circuit(n, local_k)
    flat = torch.cat([flat, torch.full((B, n - Nmax), -LARGE_NUMBER)], dim=1)          # :96-107
    marg = log_marginals(flat, c).exp()                                                # :127
    with torch.no_grad():                                                              # simple.py:107-118,244-251
        keys = flat + (-torch.log(-torch.log(uniform)))
        idx = keys.topk(local_k, dim=-1).indices
        hot = torch.zeros_like(keys).scatter_(2, idx, 1.0)
    samples = (hot - marg[None]).detach() + marg[None]                                 # :130
    samples, marg = samples[..., :Nmax], marg[:, :Nmax]                                # :133-134
    return samples.reshape(1, B, 1, Nmax).permute(0, 1, 3, 2), marg.reshape(B, 1, Nmax).permute(0, 2, 1)
