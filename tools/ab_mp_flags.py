#!/usr/bin/env python3
"""Same-process A/B of the per-graph message-passing kernel under ISG_MP_FLAGS values (read at every launch).
python3 tools/ab_mp_flags.py 1 5"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

vals = sys.argv[1:] or ["1", "5"]
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_l, x_r = torch.randn(N, H * C, device=dev, generator=g), torch.randn(N, H * C, device=dev, generator=g)
e_proj, att = torch.randn(E, H * C, device=dev, generator=g), torch.randn(1, H, C, device=dev, generator=g)
flush = torch.empty(1 << 27, device=dev)
res, outs = {}, {}
for r in range(14):
    for v in vals:
        os.environ["ISG_MP_FLAGS"] = v
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        outs[v] = ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H)[0]
        e.record()
        torch.cuda.synchronize()
        if r >= 2:
            res.setdefault(v, []).append(s.elapsed_time(e) * 1e3)
b = ops.mp_algorithmic_bytes(N, E, H, C, False)
for v in vals:
    t = sorted(res[v])
    print(f"ISG_MP_FLAGS={v}: median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us -> {b / t[len(t) // 2] / 1e3:7.1f} GB/s  "
          f"identical to first: {torch.equal(outs[v], outs[vals[0]])}")
