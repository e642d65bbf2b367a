"""Same-box A/B at BASELINE configs[1] shapes: isg_gatv2_tile_conv (one launch) against the pair it replaces
(isg_gatv2_edge_logits + isg_gatv2_mp_fwd_logits).  HIP events, interleaved rounds, a 512 MiB write between launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isubgvqa_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs})
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(1)
x_lr = torch.randn(N, 2 * H * C, device=dev, generator=g)
x_l, x_r = x_lr[:, :H * C], x_lr[:, H * C:]
w = torch.randn(H * C, C, device=dev, generator=g) * 0.1
att = torch.randn(1, H, C, device=dev, generator=g)
bias = torch.randn(H * C, device=dev, generator=g)
flush = torch.empty(1 << 27, device=dev)
tile_ptr, ntiles, cap, _ = plan.tiles(64, 256)
print(f"N={N} E={E} graphs={graphs} tiles={int(ntiles.item())} (capacity {cap}); max nodes {plan.nmax}, max edges {plan.emax}")
tile = lambda: ops.gatv2_tile_conv(x_l, x_r, wl.edge_attr, w, att, plan, H, bias=bias, want_rowmax=True)
pair = lambda: ops.gatv2_mp_edge_logits(x_l, x_r, wl.edge_attr, w, att, plan, H, bias=bias, want_rowmax=True)


def timed(fn, r):
    flush.fill_(float(r))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3


with torch.no_grad():
    a, b = tile(), pair()
    print("equal out:", torch.equal(a[0], b[0]), " equal alpha:", torch.equal(a[1], b[1]))
    tt, tp = [], []
    for r in range(23):
        x, y = timed(tile, r), timed(pair, r)
        if r >= 3:
            tt.append(x)
            tp.append(y)
byt = ops.mp_algorithmic_bytes(N, E, H, C, False)
print(f"tile conv        : {sum(tt) / len(tt):8.1f} us  (min {min(tt):.1f})  bytes_mp {byt / 1e6:.0f} MB -> {byt / (sum(tt) / len(tt)) / 1e3:.0f} GB/s")
print(f"edge-logits pair : {sum(tp) / len(tp):8.1f} us  (min {min(tp):.1f})  {byt / (sum(tp) / len(tp)) / 1e3:.0f} GB/s")
