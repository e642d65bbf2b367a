#!/usr/bin/env python3
"""Sweep the per-graph kernel's LDS window (ISG_MP_LDS_KB = x_l rows kept in LDS, i.e. workgroups per CU) on the
BASELINE configs[1] batch, interleaved in one process.   python3 tools/sweep_mp_lds.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_l, x_r = torch.randn(N, H * C, device=dev, generator=g), torch.randn(N, H * C, device=dev, generator=g)
e_proj, att = torch.randn(E, H * C, device=dev, generator=g), torch.randn(1, H, C, device=dev, generator=g)
flush = torch.empty(1 << 27, device=dev)
res = {}
settings = ["24", "32", "40", "52", "64", "80"]
for r in range(12):
    for kb in settings:
        os.environ["ISG_MP_LDS_KB"] = kb
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H)
        e.record()
        torch.cuda.synchronize()
        if r >= 2:
            res.setdefault(kb, []).append(s.elapsed_time(e) * 1e3)
b = ops.mp_algorithmic_bytes(N, E, H, C, False)
for kb in settings:
    v = sorted(res[kb])
    print(f"LDS window {kb:>3s} KB: median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us  -> {b / v[len(v) // 2] / 1e3:7.1f} GB/s")
