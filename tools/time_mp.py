#!/usr/bin/env python3
"""A/B timing of the message-passing kernels alone on the BASELINE configs[1] batch (HIP events, interleaved rounds
in one process: cdna_hip_programming.md §5.4 rule 24).   python3 tools/time_mp.py [graphs] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs})
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_l = torch.randn(N, H * C, device=dev, generator=g)
x_r = torch.randn(N, H * C, device=dev, generator=g)
e_proj = torch.randn(E, H * C, device=dev, generator=g)
att = torch.randn(1, H, C, device=dev, generator=g)
bias = torch.randn(H * C, device=dev, generator=g)
mask = (torch.rand(N, 1, device=dev, generator=g) > 0.7).float()
flush = torch.empty(1 << 27, device=dev)     # 512 MiB write between launches: cold caches, like the real layer loop
res = {}
for r in range(rounds + 2):
    for kern in ("graph", "graph+rowmax", "chunk"):       # "+rowmax": the inference path also writes max|out| per (node, head)
        for masked in (False, True):
            flush.fill_(float(r))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out, alpha = ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H, bias=bias, node_mask=mask if masked else None,
                                      kernel=kern.split("+")[0], want_rowmax=kern.endswith("rowmax"))
            e.record()
            torch.cuda.synchronize()
            if r >= 2:
                res.setdefault((kern, masked), []).append(s.elapsed_time(e) * 1e3)
for (kern, masked), v in sorted(res.items()):
    v = sorted(v)
    b = ops.mp_algorithmic_bytes(N, E, H, C, masked)
    med = v[len(v) // 2]
    print(f"{kern:12s} masked={int(masked)}  median {med:7.1f} us  min {v[0]:7.1f} us  ->  {b / med / 1e3:7.1f} GB/s "
          f"({b / med / 1e3 / 8000:.3f} of 8 TB/s)")
