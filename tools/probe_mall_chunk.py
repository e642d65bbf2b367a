#!/usr/bin/env python3
"""Would running lin_edge and the C = 300 message-passing kernel CHUNK BY CHUNK (a quarter of the graphs at a time, so that the
quarter's e_proj -- 246 MB -- is still in the 256 MB Infinity Cache when the message-passing kernel reads it) pay?  Times, per
quarter of a 4096-graph batch: the engine's lin_edge + the flat message-passing kernel back to back (warm) against the same two
after a 512 MB flush between them (what a whole-batch e_proj of 984 MB amounts to), and the whole batch in one go."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

dev = torch.device("cuda:0")
H, C = 4, 300
g = torch.Generator(device=dev).manual_seed(0)
flush = torch.empty(1 << 27, device=dev)


def setup(graphs):
    cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs})
    wl = synthetic.make_workload(cfg).to(dev)
    N, E = wl.x.size(0), wl.edge_index.size(1)
    plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
    x_l, x_r = torch.randn(N, H * C, device=dev, generator=g), torch.randn(N, H * C, device=dev, generator=g)
    ea = torch.randn(E, C, device=dev, generator=g)
    w = torch.randn(H * C, C, device=dev, generator=g) / C ** 0.5
    att = torch.randn(1, H, C, device=dev, generator=g)
    return plan, x_l, x_r, ea, w, att, N, E


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3, r


for graphs in (1024, 4096):
    plan, x_l, x_r, ea, w, att, N, E = setup(graphs)
    res = {"gemm": [], "mp_warm": [], "mp_cold": []}
    for r in range(8):
        flush.fill_(float(r))
        tg, ep = timed(lambda: ops.linear(ea, w, None))
        tw, _ = timed(lambda: ops.gatv2_mp(x_l, x_r, ep, att, plan, H, want_planes=True))
        flush.fill_(float(r) + 0.5)
        tc, _ = timed(lambda: ops.gatv2_mp(x_l, x_r, ep, att, plan, H, want_planes=True))
        if r >= 2:
            res["gemm"].append(tg); res["mp_warm"].append(tw); res["mp_cold"].append(tc)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{graphs} graphs (N={N}, E={E}, e_proj {E * H * C * 4 / 1e6:.0f} MB): lin_edge {med['gemm']:.1f} us, message passing right "
          f"behind it {med['mp_warm']:.1f} us, after a 512 MB flush {med['mp_cold']:.1f} us")
