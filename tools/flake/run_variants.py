#!/usr/bin/env python3
"""GPU side of the experiment of DESIGN.md 16.1: every variant library of tools/flake/make_variants.py launches isg_gatv2_tile_conv
on the SAME inputs over and over (round 4's failing shape: 700 graphs of 8..33 nodes with hubs, 4 heads, C = K = 128); an output
that differs from the variant's own first launch is a wrong sum.  Per variant: launches, wrong launches, and for the first wrong
values which in-edge term is missing (edge-id order), its place in the (odd first, then pairs) walk, lane and component.
  python3 tools/flake/run_variants.py [launches per input set] [input sets] > gpurun_out/flake_variants.json"""
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from isubgvqa_amd import _lib, ops
from isubgvqa_amd.models.layers import GlorotLinear
from test_gpu_ops import _rand_graphs

nums = [a for a in sys.argv[1:] if a.isdigit()]
launches = int(nums[0]) if nums else 400
nsets = int(nums[1]) if len(nums) > 1 else 4
dev = torch.device("cuda:0")
ops.GEMM_KERNEL = "panel"
H, C, K = 4, 128, 128
gen = torch.Generator().manual_seed(23)
torch.manual_seed(5)
lin_l, lin_r = GlorotLinear(128, H * C, bias=True).to(dev), GlorotLinear(128, H * C, bias=True).to(dev)
shipped = _lib.load()
sets = []
for it in range(nsets):
    sizes = torch.randint(8, 34, (700,), generator=gen).tolist()
    batch, ei = _rand_graphs(gen, sizes, extra_per_node=1.5, hub=(7, 60))
    N, E, B = batch.numel(), ei.size(1), len(sizes)
    x = (torch.randn(N, 128, generator=gen) * torch.rand(N, 1, generator=gen).mul(3).exp()).to(dev)
    ea = torch.randn(E, K, generator=gen).to(dev)
    w = (torch.randn(H * C, K, generator=gen) * 0.1).to(dev)
    att, bias = torch.randn(1, H, C, generator=gen).to(dev), torch.randn(H * C, generator=gen).to(dev)
    plan = ops.GraphPlan.build(batch.to(dev), ei.to(dev), num_graphs=B)
    with torch.no_grad():
        x_l, x_r = ops.linear_fused(x, (lin_l, lin_r))
    sets.append(dict(batch=batch, ei=ei, sizes=sizes, ea=ea, w=w, att=att, bias=bias, plan=plan, x_l=x_l, x_r=x_r))
torch.cuda.synchronize()


def use(path):
    if path is None:
        _lib._lib = shipped
        return
    variant = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(variant, name)
        fn.restype, fn.argtypes = res, args
    _lib._lib = variant


def conv(s):
    with torch.no_grad():
        return ops.gatv2_tile_conv(s["x_l"], s["x_r"], s["ea"], s["w"], s["att"], s["plan"], H, bias=s["bias"], want_rowmax=True)


def explain(s, right, wrong, alpha):
    """first differing row: which single in-edge term's removal reproduces the wrong values (per differing column)"""
    d = right != wrong
    rows = d.any(1).nonzero().flatten()
    r = int(rows[0])
    cols = d[r].nonzero().flatten()
    ei = s["ei"]
    eids = (ei[1] == r).nonzero().flatten()
    deg = eids.numel()
    hd = int(cols[0]) // C
    al = alpha[eids.to(dev), hd].double().cpu()
    xs = s["x_l"][ei[0][eids].to(dev)][:, cols].double().cpu()
    terms = al[:, None] * xs
    b = s["bias"][cols].double().cpu()
    rgt, wrg = right[r, cols].double().cpu() - b, wrong[r, cols].double().cpu() - b
    scale = terms.abs().sum(0).clamp_min(1e-30)
    fits = []
    for j in range(deg):
        fits.append(float((((terms.sum(0) - terms[j]) - wrg).abs() / scale).max()))
    j = min(range(deg), key=lambda q: fits[q]) if deg else -1
    first_pair = deg & 1                     # the walk: one odd slot first (if the in-degree is odd), then pairs
    place = "single" if (deg & 1 and j == 0) else ("first of a pair" if (j - first_pair) % 2 == 0 else "second of a pair")
    return dict(rows_differing=int(rows.numel()), row=r, in_degree=deg, head=hd, columns=cols.tolist()[:40],
                lanes=sorted({(int(c) % C) // 4 for c in cols}), components=sorted({int(c) % 4 for c in cols}),
                wrong_minus_right=(wrg - rgt).tolist()[:8], missing_term=j, missing_term_fit=fits[j] if deg else None,
                missing_term_place=place, sum_of_terms_vs_right=float(((terms.sum(0) - rgt).abs() / scale).max()))


report = {"device": torch.cuda.get_device_name(0), "launches_per_set": launches, "sets": nsets, "variants": {}}
rolled = os.path.join(ROOT, "tools", "_build", "libisg_agg_rolled.so")
use(rolled if os.path.exists(rolled) else None)
reference = [conv(s) for s in sets]
torch.cuda.synchronize()
libs = sorted(glob.glob(os.path.join(ROOT, "tools", "_build", "libisg_agg_*.so")))
order = ["unroll2", "pairs", "two_acc", "nops", "opaque_w", "scalar", "between_nop", "unroll2_no_inflight", "rolled"]
if "--shipped-only" in sys.argv:
    libs = []
libs.append(None)                # the shipped library, last
vname = lambda p: "shipped" if p is None else os.path.basename(p)[len("libisg_agg_"):-3]
libs.sort(key=lambda p: order.index(vname(p)) if vname(p) in order else 99)
for lib in libs:
    name = vname(lib)
    use(lib)
    t0 = time.time()
    wrong_launches, records, vs_rolled = 0, [], None
    for si, s in enumerate(sets):
        first = conv(s)
        torch.cuda.synchronize()
        if vs_rolled is None:
            vs_rolled = bool(torch.equal(first[0], reference[si][0]))      # two_acc rounds differently by design
        # the variant's own majority value: three launches agreeing
        for _ in range(3):
            again = conv(s)
            if not torch.equal(again[0], first[0]):
                first = again
        for li in range(launches):
            out = conv(s)
            if not torch.equal(out[0], first[0]):
                wrong_launches += 1
                if len(records) < 12:
                    rec = explain(s, first[0], out[0], out[1])
                    rec.update(set=si, launch=li)
                    records.append(rec)
    torch.cuda.synchronize()
    report["variants"][name] = dict(launches=launches * nsets, wrong_launches=wrong_launches, first_launch_equals_rolled=vs_rolled,
                                    seconds=round(time.time() - t0, 1), records=records)
    print(f"[flake] {name}: {wrong_launches} wrong of {launches * nsets} launches ({time.time() - t0:.1f} s)", file=sys.stderr, flush=True)
use(None)
print(json.dumps(report, indent=1))
