#!/usr/bin/env python3
"""Same-process, interleaved A/B of TWO BUILDS of libisg_hip.so on the whole BASELINE configs[1] step (with --full on the
full C = 300 model, with --cfg5 [--fp16] on configs[4]'s skewed batch): the shipped library against a variant built elsewhere (e.g. one .hip recompiled with a change and linked
with the shipped objects).  Box-to-box and process-to-process scatter is +-5 %; inside one process the two builds alternate
round by round.      python3 tools/ab_libs.py tools/_build/libisg_variant.so [--full] [--graphs 4096]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from isubgvqa_amd import _lib, ops, synthetic

variant_path = [a for a in sys.argv[1:] if a.endswith(".so")][0]
graphs = int(sys.argv[sys.argv.index("--graphs") + 1]) if "--graphs" in sys.argv else 4096
dev = torch.device("cuda:0")
shipped = _lib.load()
variant = ctypes.CDLL(os.path.join(ROOT, variant_path) if not os.path.isabs(variant_path) else variant_path)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(variant, name)
    fn.restype, fn.argtypes = res, args
assert variant.isg_abi_version() == _lib.ABI_VERSION

if "--full" in sys.argv:
    from isubgvqa_amd.models import build_model
    torch.manual_seed(0)
    model = build_model(synthetic.full_model_args(), None).to(dev).eval()
    wl = synthetic.make_full_workload(graphs).to(dev)
    sg = wl.scene_graphs()
    step = lambda i: model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)[0]
elif "--cfg5" in sys.argv:      # BASELINE configs[4]'s skewed generator (2048 graphs of 8-200 nodes, AIMLE): the per-graph kernels
    cfg = synthetic.WorkloadConfig(**{**synthetic.CFG5.__dict__, "num_graphs": 2048 if "--graphs" not in sys.argv else graphs,
                                      "feature_dtype": "fp16" if "--fp16" in sys.argv else "fp32"})
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    step = lambda i: model(wl, seed=1000 + i)[0]
else:
    cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs})
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    step = lambda i: model(wl, seed=1000 + i)[0]


def run(lib, n):
    _lib._lib = lib              # (derived weights are data, the same for both builds: the caches stay)
    torch.cuda.synchronize()
    for i in range(3):
        out = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


with torch.no_grad():
    res = {"shipped": [], "variant": []}
    outs = {}
    for r in range(9):            # the first two rounds of each build warm its kernels up and are dropped
        for name, lib in (("shipped", shipped), ("variant", variant)):
            ms, out = run(lib, 20 if "--full" not in sys.argv else 5)
            res[name].append(ms)
            outs[name] = out
    _lib._lib = shipped
same = torch.equal(outs["shipped"], outs["variant"])
for name, v in res.items():
    v = v[2:]
    print(f"{name:8s}: median {sorted(v)[len(v) // 2]:.4f} ms/step   rounds {' '.join(f'{x:.4f}' for x in v)}")
print("outputs of the last step", "bit-identical" if same else f"differ: max |d| = {(outs['shipped'] - outs['variant']).abs().max().item():.3e}")
