#!/usr/bin/env python3
"""Message-passing kernel on the skewed BASELINE configs[4] shape (8-200 nodes, power-law in-degree): load-balance check."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG5.__dict__, "num_graphs": graphs})
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
deg = torch.bincount(wl.edge_index[1], minlength=N)
print(f"N={N} E={E} max nodes/graph={wl.max_nodes} max edges/graph={wl.max_edges} max in-degree={int(deg.max())} "
      f"mean in-degree={float(deg.float().mean()):.2f}")
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_l = torch.randn(N, H * C, device=dev, generator=g)
x_r = torch.randn(N, H * C, device=dev, generator=g)
e_proj = torch.randn(E, H * C, device=dev, generator=g)
att = torch.randn(1, H, C, device=dev, generator=g)
flush = torch.empty(1 << 27, device=dev)
for kern in ("graph", "chunk"):
    ts = []
    for r in range(8):
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H, kernel=kern)
        e.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(s.elapsed_time(e) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    b = ops.mp_algorithmic_bytes(N, E, H, C, False)
    print(f"{kern:6s} {t:8.1f} us  {b / t / 1e3:7.1f} GB/s ({b / t / 1e3 / 8000:.3f} of 8 TB/s)")
