#!/usr/bin/env python3
"""lin_edge + message passing at configs[1]: the un-fused pair (isg_linear_panel -> isg_gatv2_mp_fwd) against
isg_gatv2_mp_fused_edge_fwd (e_proj formed on the matrix cores inside the per-graph kernel).  HIP events, interleaved."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), 4, 128
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_l, x_r = torch.randn(N, H * C, device=dev, generator=g), torch.randn(N, H * C, device=dev, generator=g)
w = torch.randn(H * C, C, device=dev, generator=g) / C ** 0.5
att = torch.randn(1, H, C, device=dev, generator=g)
ea = wl.edge_attr
res = {"linear": [], "mp": [], "fused": []}
for r in range(12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record()
    e_proj = ops.linear(ea, w)
    ev[1].record()
    out_u, al_u = ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H)
    ev[2].record()
    ev[3].record()
    out_f, al_f = ops.gatv2_mp_fused_edge(x_l, x_r, ea, w, att, plan, H)
    ev[4].record()
    torch.cuda.synchronize()
    if r >= 2:
        res["linear"].append(ev[0].elapsed_time(ev[1]) * 1e3)
        res["mp"].append(ev[1].elapsed_time(ev[2]) * 1e3)
        res["fused"].append(ev[3].elapsed_time(ev[4]) * 1e3)
med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
b = ops.mp_algorithmic_bytes(N, E, H, C, False)
print(f"lin_edge {med['linear']:.1f} us + message passing {med['mp']:.1f} us = {med['linear'] + med['mp']:.1f} us;  "
      f"fused {med['fused']:.1f} us ({b / med['fused'] / 1e3:.0f} GB/s of the un-fused algorithmic bytes);  "
      f"max |out diff| {float((out_f - out_u).abs().max()):.2e}, max |alpha diff| {float((al_f - al_u).abs().max()):.2e}")
