#!/usr/bin/env python3
"""The configs[1] step with the tile plan's in-edge cap at 256 (the default: up to four 64-slot chunks per tile, ~2.6 on average)
and at 128 / 192 (at most two / three chunks, more and smaller tiles): interleaved rounds in one process.
  python3 tools/sweep_tile_edge_cap.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

dev = torch.device("cuda:0")
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": 4096})
wl = synthetic.make_workload(cfg).to(dev)
model = synthetic.build_answer_model(cfg).to(dev).eval()
res, outs, tiles = {}, {}, {}
with torch.no_grad():
    for r in range(7):
        for cap in (256, 192, 128):
            ops.TILE_CONV_EDGES = cap
            for i in range(3):
                out = model(wl, seed=i)[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                out = model(wl, seed=i)[0]
            torch.cuda.synchronize()
            res.setdefault(cap, []).append((time.perf_counter() - t0) / 20 * 1e3)
            outs[cap] = out
            plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=4096, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
            tiles[cap] = int(plan.tiles(64, cap)[1].item())
for cap, v in res.items():
    v = sorted(v[2:])
    print(f"edge cap {cap}: {tiles[cap]} tiles, median {v[len(v) // 2]:.4f} ms/step (rounds {' '.join(f'{x:.4f}' for x in v)}); "
          f"logits equal to cap 256: {torch.equal(outs[cap], outs[256])}, max |d| {(outs[cap] - outs[256]).abs().max().item():.2e}")
