#!/usr/bin/env python3
"""Same inputs, same seed, two forwards: every result must be bit-identical.  Walks the paths of the library (tile kernels,
per-graph kernels, the split forward, configs[4]'s skewed batch with fp32 and half rows, the full C = 300 model) on fresh random
batches -- the check that found the one-launch-in-fifteen mismatch of isg_gatv2_tile_conv's unrolled aggregation loop.
  python3 tools/stress_determinism.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
ops.MIXED_MAX_FRACTION, ops.MIXED_MIN_NODES = 0.9, 0


def same(a, b):
    if a is None or b is None:
        return a is b
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return torch.equal(a, b)


def run(name, make, n, setup=None):
    bad = 0
    for it in range(n):
        model, call = make(it)
        if setup:
            setup()
        with torch.no_grad():
            r0 = call()
            r1 = call()
        torch.cuda.synchronize()
        if not same(r0, r1):
            bad += 1
            d = [(x.float() - y.float()).abs().max().item() for x, y in zip(r0, r1) if isinstance(x, torch.Tensor)]
            print(f"   {name} iteration {it}: two forwards differ, max |d| per result {d}", flush=True)
    print(f"{name}: {bad} of {n} iterations with a difference", flush=True)
    return bad


def answer(cfg_kw, seed0):
    def make(it):
        cfg = synthetic.WorkloadConfig(**{**cfg_kw, "seed": seed0 + it})
        wl = synthetic.make_workload(cfg).to(dev)
        m = synthetic.build_answer_model(cfg, weight_seed=it).to(dev).eval()
        return m, (lambda: m(wl, seed=7 + it))
    return make


def flags(**kw):
    def f():
        for k, v in kw.items():
            setattr(ops, k, v)
    return f


total = 0
base = {**synthetic.CFG2.__dict__, "num_graphs": 700}
total += run("tile kernels (configs[1] shape, 700 graphs)", answer(base, 100), iters, flags(MIXED_DISPATCH=True, FUSE_LAYER_CONV=True))
total += run("tile_conv instead of layer_conv", answer(base, 200), iters, flags(FUSE_LAYER_CONV=False))
total += run("per-graph kernels (pair + un-fused tail)", answer(base, 300), iters,
             flags(FUSE_LAYER_CONV=True, FUSE_TILE_CONV=False, FUSE_DENSE_TAIL=False, FUSE_READOUT=False))
ops.FUSE_TILE_CONV = ops.FUSE_DENSE_TAIL = ops.FUSE_READOUT = True
sizes = lambda it: tuple([20] * 300 + [100 + 7 * (it % 10)] + [20] * 200 + [70] + [20] * 100)
total += run("split forward (two streams)", lambda it: answer({**base, "num_graphs": 602, "sizes": sizes(it)}, 400)(it), iters,
             flags(SPLIT_FORWARD=True, SPLIT_STREAM=True))
c5 = {**synthetic.CFG5.__dict__, "num_graphs": 512}
total += run("configs[4] generator, fp32 rows", answer(c5, 500), iters, flags(MIXED_DISPATCH=False))
total += run("configs[4] generator, half rows", answer({**c5, "feature_dtype": "fp16"}, 600), iters)
ops.MIXED_DISPATCH = True


def full(it):
    from isubgvqa_amd.models import build_model
    torch.manual_seed(it)
    m = build_model(synthetic.full_model_args(), None).to(dev).eval()
    wl = synthetic.make_full_workload(256, seed=900 + it) if "seed" in synthetic.make_full_workload.__code__.co_varnames \
        else synthetic.make_full_workload(256)
    wl = wl.to(dev)
    sg = wl.scene_graphs()
    return m, (lambda: m(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)[:3])


total += run("full model (C = 300, 256 graphs)", full, max(iters // 6, 4))


def full_small(it):       # round 6: the latency regime (isg_linear_skinny, the one-launch plan, un-fused wide layers), 1-12 questions
    from isubgvqa_amd.models import build_model
    torch.manual_seed(it)
    m = build_model(synthetic.full_model_args(text_vocab_size=4096), None).to(dev).eval()
    wl = synthetic.make_full_workload(1 + it % 12, tokens=6 + it % 9, seed=1900 + it, text_vocab=4096).to(dev)
    sg = wl.scene_graphs()
    return m, (lambda: m(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)[:3])


total += run("full model at 1-12 questions (small-batch kernels)", full_small, max(iters // 2, 8))
print("TOTAL differences:", total)
sys.exit(1 if total else 0)
