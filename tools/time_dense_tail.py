"""Same-box A/B at BASELINE configs[1] shapes: isg_mgat_dense_tail (x_proj + layer tail + next instruction gate on graph-aligned
tiles) against the un-fused chain it replaces (2 x isg_linear_f16x3_tile, isg_instr_attn_graphnorm_residual, isg_instr_gate).
HIP events, interleaved rounds, a 512 MiB write between launches (cold caches, as between the kernels of a step)."""
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
m = net.gat_seq
N, H, C = wl.x.size(0), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(1)
conv_out = torch.randn(N, H * C, device=dev, generator=g)
rm = conv_out.view(N, H, C).abs().amax(dim=2).contiguous()
h = torch.randn(N, C, device=dev, generator=g)
ins, ins_next = wl.instr[0].contiguous(), wl.instr[1].contiguous()
bn = m.bns[0]
flush = torch.empty(1 << 27, device=dev)
tile_ptr, ntiles, cap, _ = plan.tiles(64)
print(f"N={N} graphs={graphs} tiles={int(ntiles.item())} (capacity {cap}), rows/tile={N / max(int(ntiles.item()), 1):.1f}")


def fused():
    ops.attach_row_maxima(conv_out, rm)
    return ops.mgat_dense_tail(conv_out, m.x_proj[0], ins, h, plan, bn.weight, bn.bias, bn.mean_scale, bn.eps, ins_next=ins_next)[:2]


def chain():
    ops.attach_row_maxima(conv_out, rm)
    c = ops.mlp(m.x_proj[0], conv_out)
    hh = ops.mgat_layer_tail(ins, c, h, plan, bn.weight, bn.bias, bn.mean_scale, bn.eps)
    return hh, ops.instr_gate(hh, ins_next, wl.batch, plan=plan)


def timed(fn, r):
    flush.fill_(float(r))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3


with torch.no_grad():
    a, b = fused(), chain()
    print("max |h fused - h chain| =", (a[0] - b[0]).abs().max().item(), " max |xg diff| =", (a[1] - b[1]).abs().max().item())
    tf, tc = [], []
    for r in range(23):
        x, y = timed(fused, r), timed(chain, r)
        if r >= 3:
            tf.append(x)
            tc.append(y)
flops = 2.0 * N * (512 * 256 + 256 * 128) * 3
print(f"fused dense tail : {sum(tf) / len(tf):8.1f} us  (min {min(tf):.1f})  {flops / (sum(tf) / len(tf)) / 1e6:.0f} TF/s of fp16 products")
print(f"un-fused chain   : {sum(tc) / len(tc):8.1f} us  (min {min(tc):.1f})")
