#!/usr/bin/env python3
"""BASELINE configs[4]'s generator at C = 128 with fp32 rows: mixed dispatch on / off, ms per step.
  python3 tools/time_cfg5.py [graphs] [--off] [--order=desc|asc]   (under rocprofv3 for the per-kernel split)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

args = [a for a in sys.argv[1:] if not a.startswith("--")]
graphs = int(args[0]) if args else 2048
dev = torch.device("cuda:0")
ops.MIXED_MAX_FRACTION, ops.MIXED_MIN_NODES = 0.9, 0      # the gate open: this tool measures the mode itself
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG5.__dict__, "num_graphs": graphs,
                                  "feature_dtype": "fp16" if "--fp16" in sys.argv else "fp32"})       # --fp16: configs[4]'s half rows
order = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--order=")]       # the same graphs, largest / smallest first
if order:
    sizes = sorted(synthetic.graph_sizes(cfg, torch.Generator().manual_seed(cfg.seed)).tolist(), reverse=order[0] == "desc")
    cfg = synthetic.WorkloadConfig(**{**cfg.__dict__, "sizes": tuple(sizes)})
    print(f"graphs ordered by size ({order[0]}): {sizes[:3]} ... {sizes[-3:]}")
wl = synthetic.make_workload(cfg).to(dev)
model = synthetic.build_answer_model(cfg).to(dev).eval()
for mode in ([False] if "--off" in sys.argv else [True] if "--on" in sys.argv else [True, False, True, False]):
    ops.MIXED_DISPATCH = mode
    with torch.no_grad():
        for i in range(3):
            model(wl, seed=50 + i)
        torch.cuda.synchronize()
        ops.reset_counters()
        t0 = time.perf_counter()
        for i in range(10):
            model(wl, seed=60 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    c = ops.counters()
    print(f"mixed dispatch {'on ' if mode else 'off'}: {dt * 1e3:.3f} ms/step, N={wl.x.size(0)}, tile node visits {c['tile_nodes']}, "
          f"per-graph {c['oversize_nodes']}", flush=True)
