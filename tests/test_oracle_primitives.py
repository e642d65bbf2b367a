"""Known-answer and brute-force pins for the restated third-party primitives
(oracle/primitives.py).  CPU only."""
import math

import torch

from oracle import model as OM
from oracle import primitives as P


def test_scatter_known_answers():
    src = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]])
    idx = torch.tensor([2, 0, 2, 2])
    assert torch.equal(P.scatter_sum(src, idx, 4), torch.tensor([[3.0, 4.0], [0, 0], [13.0, 16.0], [0, 0]]))
    m = P.scatter_mean(src, idx, 4)
    assert torch.allclose(m, torch.tensor([[3.0, 4.0], [0, 0], [13.0 / 3, 16.0 / 3], [0, 0]]))
    assert torch.equal(P.scatter_max(src, idx, 4), torch.tensor([[3.0, 4.0], [0, 0], [7.0, 8.0], [0, 0]]))


def test_pyg_softmax_known_answer_and_epsilon():
    src = torch.tensor([0.0, math.log(3.0), 5.0])
    idx = torch.tensor([1, 1, 0])
    out = P.pyg_softmax(src, idx, 3)
    # group 1: exp(0-log3)=1/3, exp(0)=1 -> sum 4/3 (+1e-16)
    assert torch.allclose(out, torch.tensor([0.25, 0.75, 1.0]), atol=1e-7)
    out2 = P.scatter_softmax_1d(src, idx, 3)
    assert torch.allclose(out2, torch.tensor([0.25, 0.75, 1.0]), atol=1e-7)


def test_graph_norm_matches_per_graph_formula():
    torch.manual_seed(0)
    x = torch.randn(9, 5)
    batch = torch.tensor([0, 0, 0, 1, 1, 2, 2, 2, 2])
    w, b, ms = torch.randn(5), torch.randn(5), torch.randn(5)
    y = P.graph_norm(x, batch, w, b, ms)
    for g in range(3):
        xs = x[batch == g].double()
        mean = xs.mean(0)
        o = xs - mean * ms.double()
        var = (o * o).mean(0)
        ref = w.double() * o / (var + 1e-5).sqrt() + b.double()
        assert torch.allclose(y[batch == g].double(), ref, atol=1e-5)


def test_to_dense_batch_zero_fill_and_order():
    x = torch.arange(1.0, 7.0).view(6, 1)
    batch = torch.tensor([0, 0, 1, 2, 2, 2])
    d, m = P.to_dense_batch(x, batch)
    assert d.shape == (3, 3, 1)
    assert torch.equal(d.squeeze(-1), torch.tensor([[1.0, 2, 0], [3, 0, 0], [4, 5, 6]]))
    assert torch.equal(m, torch.tensor([[True, True, False], [True, False, False], [True, True, True]]))


def _brute_force_mp(x_l, x_r, e_proj, att, ei, em, slope):
    """Independent per-destination dense loops in float64 (no scatter primitives)."""
    N, H, C = x_l.shape
    E = ei.size(1)
    out = torch.zeros(N, H, C, dtype=torch.float64)
    alpha = torch.zeros(E, H, dtype=torch.float64)
    xl, xr, ep, at = x_l.double(), x_r.double(), e_proj.double(), att.double().view(H, C)
    for i in range(N):
        inc = [e for e in range(E) if int(ei[1, e]) == i]
        if not inc:
            continue
        for h in range(H):
            logits = []
            for e in inc:
                s = xr[i, h] + xl[int(ei[0, e]), h] + ep[e, h]
                m = 1.0 if em is None else float(em[e])
                s = s * m
                s = torch.where(s > 0, s, s * slope)
                s = s * m
                logits.append(float((s * at[h]).sum()))
            mx = max(logits)
            ex = [math.exp(l - mx) for l in logits]
            den = sum(ex) + 1e-16
            for e, v in zip(inc, ex):
                a = v / den
                alpha[e, h] = a
                m = 1.0 if em is None else float(em[e])
                out[i, h] += xl[int(ei[0, e]), h] * a * m
    return out, alpha


def test_message_passing_against_dense_brute_force():
    torch.manual_seed(3)
    N, H, C = 11, 4, 6
    src = torch.randint(0, N, (40,))
    dst = torch.randint(0, N - 1, (40,))          # node N-1 is an isolated target
    ei = torch.stack([src, dst])
    x_l, x_r = torch.randn(N, H, C), torch.randn(N, H, C)
    e_proj = torch.randn(40, H, C)
    att = torch.randn(1, H, C)
    for em in (None, (torch.rand(40, 1) > 0.4).float()):
        out, alpha = OM.gatv2_message_passing(x_l, x_r, e_proj, att, ei, em, 0.2)
        bo, ba = _brute_force_mp(x_l, x_r, e_proj, att, ei, em, 0.2)
        assert torch.allclose(out.double(), bo, atol=1e-5)
        assert torch.allclose(alpha.double(), ba, atol=1e-6)
        assert torch.equal(out[N - 1], torch.zeros(H, C))
