#!/usr/bin/env python3
"""Message-passing kernels at the reference's default width (H = 4, C = 300: 1200-float rows) on the configs[1] topology.
python3 tools/time_mp_c300.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), 4, 300
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_l, x_r = torch.randn(N, H * C, device=dev, generator=g), torch.randn(N, H * C, device=dev, generator=g)
e_proj, att = torch.randn(E, H * C, device=dev, generator=g), torch.randn(1, H, C, device=dev, generator=g)
flush = torch.empty(1 << 27, device=dev)
variants = [("graph", {}), ("chunk", {})] + [("graph", {"ISG_MP_HS": hs, "ISG_MP_LDS_KB": kb}) for hs, kb in (("2", "40"), ("2", "64"), ("4", "64"))
                                             if os.environ.get("ISG_MP_TRY_HS")]
res, outs = {}, {}
for r in range(10):
    for i, (kern, env) in enumerate(variants):
        for k in ("ISG_MP_HS", "ISG_MP_LDS_KB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        outs[i] = ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H, kernel=kern)[0]
        e.record()
        torch.cuda.synchronize()
        if r >= 2:
            res.setdefault(i, []).append(s.elapsed_time(e) * 1e3)
b = ops.mp_algorithmic_bytes(N, E, H, C, False)
for i, (kern, env) in enumerate(variants):
    t = sorted(res[i])
    print(f"{kern:6s} {env}: median {t[len(t) // 2]:7.1f} us -> {b / t[len(t) // 2] / 1e3:7.1f} GB/s ({b / t[len(t) // 2] / 8e6:.3f} of 8 TB/s)  "
          f"max |diff| vs first {float((outs[i] - outs[0]).abs().max()):.2e}")
