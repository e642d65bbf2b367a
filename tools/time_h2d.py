#!/usr/bin/env python3
"""PCIe-inclusive rate of the configs[1] step: the batch handed over as pinned HOST buffers (dense fp32 features, as the
synthetic workload defines them; and the int64 token batch the real model receives), copied per step.
python3 tools/time_h2d.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import synthetic

dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg)
model = synthetic.build_answer_model(cfg).to(dev).eval()
host = synthetic.Workload(*[t.pin_memory() for t in (wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.instr, wl.glf)],
                          wl.num_graphs, wl.max_nodes, wl.max_edges)
nbytes = sum(t.numel() * t.element_size() for t in (host.x, host.edge_index, host.edge_attr, host.batch, host.instr, host.glf))
copy_stream = torch.cuda.Stream()


def run(mode, steps=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nxt = None
    with torch.no_grad():
        for i in range(steps):
            if mode == "resident":
                cur = dev_wl
            elif mode == "serial":
                cur = host.to(dev)                                # default stream: copy then compute
            else:                                                 # overlapped: next batch on a copy stream
                if nxt is None:
                    with torch.cuda.stream(copy_stream):
                        nxt = synthetic.Workload(*[t.to(dev, non_blocking=True) for t in (host.x, host.edge_index, host.edge_attr, host.batch, host.instr, host.glf)], host.num_graphs, host.max_nodes, host.max_edges)
                torch.cuda.current_stream().wait_stream(copy_stream)
                cur = nxt
                with torch.cuda.stream(copy_stream):
                    nxt = synthetic.Workload(*[t.to(dev, non_blocking=True) for t in (host.x, host.edge_index, host.edge_attr, host.batch, host.instr, host.glf)], host.num_graphs, host.max_nodes, host.max_edges)
            model(cur, seed=i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


dev_wl = wl.to(dev)
for mode in ("resident", "serial", "overlapped"):
    run(mode, 5)
    dt = run(mode)
    print(f"{mode:10s}: {dt * 1e3:6.2f} ms/step = {cfg.num_graphs / dt:,.0f} questions/s")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    host.to(dev)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"batch = {nbytes / 1e6:.1f} MB of pinned host memory, H2D {ms:.2f} ms = {nbytes / ms / 1e6:.1f} GB/s")
