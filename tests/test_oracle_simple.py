"""The oracle's SIMPLE sampler against masks / marginals / gradients produced by the real
EdgeSIMPLEBatched.forward (golden G9), including rows whose zero pads outnumber k and a NaN row."""
import torch

from conftest import load_golden
from oracle import simple as OS


def test_circuit_tables_small_case():
    c = OS.build_circuit(8, 2)
    assert c.levels == 3 and c.cap == [1, 2, 2, 2]
    assert c.reach[3] == [False, False, True] and all(c.reach[0][:2])
    assert c.max_elements == 3 and c.n_elem[1][:3] == [1, 2, 1]


def test_g9_simple_matches_reference():
    for c in load_golden("g9_simple.pt"):
        th = c["scores"].clone().requires_grad_(True)
        mask, marg = OS.simple_forward(th, c["k"], c["uniform"])
        assert torch.equal(torch.isnan(marg), torch.isnan(c["marginals"]))
        torch.testing.assert_close(marg, c["marginals"], rtol=1e-5, atol=1e-6, equal_nan=True)
        torch.testing.assert_close(mask, c["mask"], rtol=0, atol=1e-6, equal_nan=True)
        (torch.nan_to_num(mask.squeeze(0)) * c["w"]).sum().backward()
        torch.testing.assert_close(th.grad, c["grad"], rtol=1e-4, atol=2e-6, equal_nan=True)
