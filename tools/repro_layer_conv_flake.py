#!/usr/bin/env python3
"""Hunt for the intermittent mismatch of isg_gatv2_layer_conv against isg_linear_f16x3 + isg_gatv2_tile_conv seen in two of nine
full `pytest -m gpu` runs (test_layer_conv_is_bit_identical_to_projection_plus_tile_conv[4-128-None], its 700-graph case):
the same comparison in a loop, every launch repeated, with the rows / heads / tiles of a mismatch printed.
  python3 tools/repro_layer_conv_flake.py [iterations] [--after-split]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from isubgvqa_amd import ops, synthetic
from isubgvqa_amd.models.layers import GlorotLinear
from test_gpu_ops import _rand_graphs

lib_arg = [a for a in sys.argv[1:] if a.endswith(".so")]
if lib_arg:             # a variant build of the library (tools/_build/...) in place of the shipped one
    import ctypes
    from isubgvqa_amd import _lib
    variant = ctypes.CDLL(os.path.abspath(lib_arg[0]))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(variant, name)
        fn.restype, fn.argtypes = res, args
    _lib.load()
    _lib._lib = variant
    print("library:", lib_arg[0])
iters = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 40
dev = torch.device("cuda:0")
if "--after-split" in sys.argv:            # the state the full suite is in when the test runs: a side stream has been used
    sizes = (20,) * 150 + (130,) + (20,) * 100
    cfg = synthetic.WorkloadConfig(num_graphs=len(sizes), sizes=sizes, sampler="imle", seed=11)
    wl = synthetic.make_workload(cfg).to(dev)
    m = synthetic.build_answer_model(cfg).eval().to(dev)
    ops.MIXED_MAX_FRACTION, ops.MIXED_MIN_NODES = 0.9, 0
    with torch.no_grad():
        for i in range(3):
            m(wl)
    torch.cuda.synchronize()
    print("split forward ran:", ops.counters()["oversize_nodes"] > 0)
ops.GEMM_KERNEL = "panel"
H, C, K = 4, 128, 128
gen = torch.Generator().manual_seed(23)
torch.manual_seed(5)
lin_l, lin_r = GlorotLinear(128, H * C, bias=True).to(dev), GlorotLinear(128, H * C, bias=True).to(dev)
bad = 0
for it in range(iters):
    sizes = torch.randint(8, 34, (700,), generator=gen).tolist()
    batch, ei = _rand_graphs(gen, sizes, extra_per_node=1.5, hub=(7, 60))
    N, E, B = batch.numel(), ei.size(1), len(sizes)
    x = (torch.randn(N, 128, generator=gen) * torch.rand(N, 1, generator=gen).mul(3).exp()).to(dev)
    ea = torch.randn(E, K, generator=gen).to(dev)
    w = (torch.randn(H * C, K, generator=gen) * 0.1).to(dev)
    att, bias = torch.randn(1, H, C, generator=gen).to(dev), torch.randn(H * C, generator=gen).to(dev)
    plan = ops.GraphPlan.build(batch.to(dev), ei.to(dev), num_graphs=B)
    with torch.no_grad():
        f = [ops.gatv2_layer_conv(x, lin_l, lin_r, ea, w, att, plan, H, bias=bias, want_rowmax=True) for _ in range(2)]
        x_l, x_r = ops.linear_fused(x, (lin_l, lin_r))
        t = [ops.gatv2_tile_conv(x_l, x_r, ea, w, att, plan, H, bias=bias, want_rowmax=True) for _ in range(2)]
    torch.cuda.synchronize()
    eq = lambda a, b: torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    if not (eq(f[0], f[1]) and eq(t[0], t[1]) and eq(f[0], t[0])):
        bad += 1
        print(f"iteration {it}: layer_conv stable {eq(f[0], f[1])}, tile_conv stable {eq(t[0], t[1])}, equal {eq(f[0], t[0])}")
        for name, a, b in (("layer#0 vs layer#1", f[0][0], f[1][0]), ("tile#0 vs tile#1", t[0][0], t[1][0]),
                           ("layer#0 vs tile#0", f[0][0], t[0][0]), ("layer#1 vs tile#1", f[1][0], t[1][0])):
            d = (a != b)
            if d.any():
                rows = d.any(1).nonzero().flatten()
                heads = d.view(N, H, C).any(2).any(0).nonzero().flatten().tolist()
                g = batch[rows.cpu()].unique().tolist()
                print(f"   {name}: {int(d.sum())} values in {rows.numel()} rows {rows[:8].tolist()}.., heads {heads}, graphs {g[:10]}, "
                      f"max |d| {(a - b).abs().max().item():.3g}")
                r = int(rows[0])
                cols = d[r].nonzero().flatten().tolist()
                gph = int(batch[r])
                first = int((batch == gph).nonzero()[0])
                srcs = ei[0][ei[1] == r].tolist()
                print(f"      row {r} = node {r - first} of graph {gph} ({sizes[gph]} nodes), columns {cols}; in-edges from {[s_ - first for s_ in srcs]}")
                print(f"      a: {a[r, cols[:4]].tolist()}  b: {b[r, cols[:4]].tolist()}")
                if name.startswith("layer#") and "--explain" in sys.argv:      # a = the layer kernel (right), b = tile_conv (wrong)
                    hd = cols[0] // C
                    eids = (ei[1] == r).nonzero().flatten()
                    al = t[0][1][eids.to(dev), hd].double().cpu()                    # alpha of the row's in-edges (edge-id order)
                    xs = x_l[ei[0][eids].to(dev)][:, cols].double().cpu()             # their x_l values at the differing columns
                    terms = al[:, None] * xs
                    right = a[r, cols].double().cpu() - bias[cols].double().cpu()
                    wrong = b[r, cols].double().cpu() - bias[cols].double().cpu()
                    print(f"      in-degree {eids.numel()}, alpha {al.tolist()}")
                    print(f"      sum of terms - right: {(terms.sum(0) - right).abs().max().item():.2e}")
                    for j in range(eids.numel()):
                        print(f"      without term {j}: max |.. - wrong| = {((terms.sum(0) - terms[j]) - wrong).abs().max().item():.3e};"
                              f"  term {j} alone: {(terms[j] - wrong).abs().max().item():.3e};"
                              f"  term {j} twice: {((terms.sum(0) + terms[j]) - wrong).abs().max().item():.3e}")
                    # the half-wave's previous node (k - 8 of the tile) and the other component pair of the same lanes
                    print(f"      wrong - right: {(wrong - right)[:6].tolist()}")
                    for dr in (-8, -1, 1, 8):
                        if 0 <= r + dr < N:
                            other = a[r + dr, cols].double().cpu() - bias[cols].double().cpu()
                            print(f"      right row {r + dr:+d}: max |wrong - that| = {(wrong - other).abs().max().item():.3e}; "
                                  f"|wrong - right - that| = {(wrong - right - other).abs().max().item():.3e}")
                    for dc in (-2, -1, 1, 2):
                        cc = [c + dc for c in cols]
                        if min(cc) >= 0 and max(cc) < H * C:
                            other = a[r, cc].double().cpu() - bias[cc].double().cpu()
                            print(f"      right columns {dc:+d}: max |wrong - that| = {(wrong - other).abs().max().item():.3e}")
print(f"{bad} of {iters} iterations with a mismatch")
