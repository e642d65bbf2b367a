#!/usr/bin/env python3
"""Where the HOST spends a step: cProfile of the model's forward (launches issued, nothing waited for) on a configs[1] batch.
  python3 tools/profile_host.py [graphs] [big graphs]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nbig = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(3)
base = synthetic.graph_sizes(synthetic.WorkloadConfig(num_graphs=graphs), gen).tolist()
for i in range(nbig):
    base[(i * 977 + 13) % graphs] = 100 + (i * 37) % 90
cfg = synthetic.WorkloadConfig(num_graphs=graphs, sizes=tuple(base))
wl = synthetic.make_workload(cfg).to(dev)
model = synthetic.build_answer_model(cfg).to(dev).eval()
with torch.no_grad():
    for i in range(5):
        model(wl, seed=50 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        model(wl, seed=60 + i)
    t_issue = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    print(f"{graphs} graphs, {nbig} beyond a tile: host issue {t_issue * 1e3:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20):
        model(wl, seed=80 + i)
        if i % 4 == 3:
            torch.cuda.synchronize()       # the queue never fills: the profile shows issue cost, not back-pressure
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(45)
