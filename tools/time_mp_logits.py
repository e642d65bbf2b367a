#!/usr/bin/env python3
"""One layer's edge projection + message passing at the BASELINE configs[1] topology, un-fused against the edge-logits
pair (HIP events, interleaved rounds in one process, 512 MiB written between launches: cold caches as in the layer loop).
    python3 tools/time_mp_logits.py [graphs] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import _lib, ops, synthetic

if os.environ.get("ISG_TOOL_LIB"):        # A/B against another build of the library (same box, separate processes)
    _lib.LIB_PATH = os.path.abspath(os.environ["ISG_TOOL_LIB"])

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
chan = int(sys.argv[3]) if len(sys.argv) > 3 else 128          # channels per head (K stays <= 128: the edge-logits kernel's limit)
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs, "channels": chan})
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
K = min(wl.edge_attr.size(1), int(sys.argv[4]) if len(sys.argv) > 4 else 128)
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_lr = torch.randn(N, 2 * H * C, device=dev, generator=g)
x_l, x_r = x_lr[:, :H * C], x_lr[:, H * C:]
ea = wl.edge_attr.float()[:, :K].contiguous()
w = torch.randn(H * C, K, device=dev, generator=g) / K ** 0.5
att = torch.randn(1, H, C, device=dev, generator=g)
bias = torch.randn(H * C, device=dev, generator=g)
mask = (torch.rand(N, 1, device=dev, generator=g) > 0.7).float()
flush = torch.empty(1 << 27, device=dev)


def timed(fn, r, key, res):
    flush.fill_(float(r))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = fn()
    e.record()
    torch.cuda.synchronize()
    if r >= 2:
        res.setdefault(key, []).append(s.elapsed_time(e) * 1e3)
    return out


res = {}
for r in range(rounds + 2):
    for masked in (False, True):
        nm = mask if masked else None
        ep = timed(lambda: ops.linear(ea, w), r, ("1 lin_edge (isg_linear_f16x3)", masked), res)
        timed(lambda: ops.gatv2_mp(x_l, x_r, ep, att, plan, H, bias=bias, node_mask=nm, want_rowmax=True), r,
              ("2 message passing, e_proj streamed", masked), res)
        lg = timed(lambda: ops.gatv2_edge_logits(x_l, x_r, ea, w, att, plan, H, node_mask=nm), r,
                   ("3 edge logits (isg_gatv2_edge_logits)", masked), res)
        timed(lambda: ops.gatv2_mp_edge_logits(x_l, x_r, ea, w, att, plan, H, bias=bias, node_mask=nm, want_rowmax=True), r,
              ("4 pair: edge logits + message passing from logits", masked), res)
print(f"N={N} E={E} H={H} C={C} K={K}")
for (name, masked), v in sorted(res.items()):
    v = sorted(v)
    print(f"{name:52s} masked={int(masked)}  median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us")
