#!/usr/bin/env python3
"""Host time per ops.* call (inclusive, no device sync inside) over a forward of AnswerModel, tile path and per-graph path:
where the ~25 us per launch go.   python3 tools/host_op_times.py"""
import collections
import functools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

dev = torch.device("cuda:0")
acc = collections.defaultdict(lambda: [0, 0.0])
depth = [0]


def wrap(name, fn):
    @functools.wraps(fn)
    def w(*a, **k):
        depth[0] += 1
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            depth[0] -= 1
            if depth[0] == 0:          # outermost ops call only: inclusive of the ops it calls
                acc[name][0] += 1
                acc[name][1] += dt
    return w


names = [n for n, v in vars(ops).items() if callable(v) and not n.startswith("_") and getattr(v, "__module__", "") == ops.__name__
         and not isinstance(v, type)]
for n in names:
    setattr(ops, n, wrap(n, getattr(ops, n)))
ops.GraphPlan.build = staticmethod(wrap("GraphPlan.build", ops.GraphPlan.build))


def run(cfg, label, steps=30, **sw):
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    with ops.configured(**sw), torch.no_grad():
        for i in range(5):
            model(wl, seed=i)
        torch.cuda.synchronize()
        acc.clear()
        t0 = time.perf_counter()
        for i in range(steps):
            model(wl, seed=10 + i)
            if i % 5 == 4:
                torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        torch.cuda.synchronize()
    inside = sum(v[1] for v in acc.values())
    print(f"== {label}: {tot / steps * 1e6:.0f} us per forward on the host, {inside / steps * 1e6:.0f} us inside ops.* ({sum(v[0] for v in acc.values()) / steps:.0f} calls)")
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"   {n:28s} {c / steps:5.1f} calls  {t / steps * 1e6:7.1f} us  ({t / c * 1e6:5.1f} us each)")


run(synthetic.WorkloadConfig(num_graphs=1024, sampler="gumbel", seed=5), "tile kernels, 1024 graphs")
sizes = (100, 130, 90, 160)
run(synthetic.WorkloadConfig(num_graphs=len(sizes), sizes=sizes, sampler="gumbel", seed=5), "per-graph kernels, 4 big graphs", mixed_dispatch=False)
