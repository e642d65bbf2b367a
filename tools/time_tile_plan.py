#!/usr/bin/env python3
"""isg_tile_plan_edge_planes (the tile plan as workgroup 0 of the launch that splits the edge rows) on the configs[1] batch:
us per launch with and without the heavy-first list.   python3 tools/time_tile_plan.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

dev = torch.device("cuda:0")
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": 4096})
wl = synthetic.make_workload(cfg).to(dev)
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=4096, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
for heavy in (True, False, True, False):
    ops.TILE_HEAVY_FIRST = heavy
    ts = []
    for r in range(30):
        plan._tiles, plan._edge_planes = None, None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        plan.tiles_and_edge_planes(wl.edge_attr, ops.TILE_CONV_NODES, ops.TILE_CONV_EDGES)
        e.record()
        torch.cuda.synchronize()
        if r >= 5:
            ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    print(f"heavy-first list {'on ' if heavy else 'off'}: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us")


def timed(fn, n=30):
    ts = []
    for r in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if r >= 5:
            ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def plan_alone():
    plan._tiles = None
    plan.tiles(ops.TILE_CONV_NODES, ops.TILE_CONV_EDGES)


def planes_alone():
    plan._edge_planes = None
    plan.edge_planes(wl.edge_attr)


print("tile plan alone (one workgroup): median %.1f us, min %.1f us" % timed(plan_alone))
print("edge planes alone:               median %.1f us, min %.1f us" % timed(planes_alone))
print("an empty launch pair (events only): median %.1f us, min %.1f us" % timed(lambda: None))
