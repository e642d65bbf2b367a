"""Same-box A/B at BASELINE configs[1] shapes: isg_gatv2_layer_conv (lin_l | lin_r + message passing, one launch) against
isg_linear_f16x3 + isg_gatv2_tile_conv (two launches) and against round 2's three (projection + edge-logits pair)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isubgvqa_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs})
wl = synthetic.make_workload(cfg).to(dev)
net = synthetic.build_answer_model(cfg).to(dev).eval()
conv = net.gat_seq.convs[0]
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
x = wl.x.contiguous()
xp = ops.node_planes(x)       # the layer kernel's input as its producers (instruction gate / previous tail) write it
flush = torch.empty(1 << 27, device=dev)
w, att, bias = conv.lin_edge.weight, conv.att, conv.bias


def fused():
    return ops.gatv2_layer_conv(xp, conv.lin_l, conv.lin_r, wl.edge_attr, w, att, plan, H, bias=bias, want_rowmax=True)


def two():
    x_l, x_r = ops.linear_fused(x, (conv.lin_l, conv.lin_r))
    return ops.gatv2_tile_conv(x_l, x_r, wl.edge_attr, w, att, plan, H, bias=bias, want_rowmax=True)


def three():
    x_l, x_r = ops.linear_fused(x, (conv.lin_l, conv.lin_r))
    return ops.gatv2_mp_edge_logits(x_l, x_r, wl.edge_attr, w, att, plan, H, bias=bias, want_rowmax=True)


def timed(fn, r):
    flush.fill_(float(r))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3


with torch.no_grad():
    a, b, c = fused(), two(), three()
    print("equal out (fused vs two launches):", torch.equal(a[0], b[0]), " vs round 2's three:", torch.equal(a[0], c[0]),
          " equal alpha:", torch.equal(a[1], b[1]))
    ts = {"fused": [], "two": [], "three": []}
    for r in range(23):
        for name, fn in (("fused", fused), ("two", two), ("three", three)):
            v = timed(fn, r)
            if r >= 3:
                ts[name].append(v)
for name, label in (("fused", "isg_gatv2_layer_conv (1 launch)"), ("two", "projection + tile conv (2 launches)"),
                    ("three", "projection + edge logits + mp (3 launches, r02)")):
    v = ts[name]
    print(f"{label:50s}: {sum(v) / len(v):8.1f} us  (min {min(v):.1f})")
