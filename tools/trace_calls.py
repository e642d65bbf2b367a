#!/usr/bin/env python3
"""The C-ABI calls (in order) and the torch ops of ONE forward of AnswerModel: (a) a configs[1]-shaped batch on the tile kernels,
(b) a sub-batch of oversize graphs on the per-graph kernels.  What a native executor for either would have to chain.
  python3 tools/trace_calls.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import _lib, ops, synthetic

dev = torch.device("cuda:0")
real = _lib.load()


class Tracer:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(real, name)

        def wrapped(*a):
            self.calls.append(name)
            return fn(*a)
        return wrapped


def run(cfg, label):
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    with torch.no_grad():
        for i in range(2):
            model(wl, seed=3 + i)
        torch.cuda.synchronize()
        tr = Tracer()
        _lib._lib = tr
        try:
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
                model(wl, seed=9)
        finally:
            _lib._lib = real
        torch.cuda.synchronize()
    aten = [e.key for e in prof.key_averages() if e.key.startswith("aten::")]
    n_aten = sum(e.count for e in prof.key_averages() if e.key.startswith("aten::"))
    print(f"== {label}: {len(tr.calls)} C-ABI calls, {n_aten} aten ops ({len(aten)} kinds)")
    print("   " + " ".join(tr.calls))
    top = sorted((e for e in prof.key_averages() if e.key.startswith("aten::")), key=lambda e: -e.count)[:14]
    print("   aten: " + ", ".join(f"{e.key[6:]} x{e.count}" for e in top))


run(synthetic.WorkloadConfig(num_graphs=256, sampler="gumbel", seed=5), "tile kernels (256 graphs of ~20 nodes)")
sizes = (100, 130, 90, 160)
with ops.configured(mixed_dispatch=False):
    run(synthetic.WorkloadConfig(num_graphs=len(sizes), sizes=sizes, sampler="gumbel", seed=5), "per-graph kernels (4 graphs of 90-160 nodes)")
