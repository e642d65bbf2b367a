#!/usr/bin/env python3
"""Where does a wave of isg_gatv2_layer_conv spend its cycles?  Uses tools/_build/libisg_dt_stamp.so (tools/stamp_dense_tail.py
--build makes it with -DISG_DIAG)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build", "libisg_dt_stamp.so")

import torch

from isubgvqa_amd import _lib, ops, synthetic

stamp = ctypes.CDLL(OUT)
stamp.isg_gatv2_layer_conv.restype, stamp.isg_gatv2_layer_conv.argtypes = _lib.SIGNATURES["isg_gatv2_layer_conv"]
stamp.isg_lc_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
net = synthetic.build_answer_model(cfg).to(dev).eval()
conv = net.gat_seq.convs[0]
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
x = wl.x.contiguous()
cat_w = torch.cat([conv.lin_l.weight.detach(), conv.lin_r.weight.detach()], 0).contiguous()
cat_b = torch.cat([conv.lin_l.bias.detach(), conv.lin_r.bias.detach()]).contiguous()
wn, wn_inv = ops._weight_planes(cat_w, False, "f16x3")
we, we_inv = ops._weight_planes(conv.lin_edge.weight, True, "f16x3")
ep, ep_inv = plan.edge_planes(wl.edge_attr)
_, ntiles, cap, tile_info = plan.tiles(64, 256)
T = int(ntiles.item())
out = torch.empty(N, H * C, device=dev)
alpha = torch.empty(E, H, device=dev)
rowmax = torch.empty(N, H, device=dev)
xp = ops.node_planes(x)
buf = torch.zeros(4096 * 8, 16, dtype=torch.int64, device=dev)
assert stamp.isg_lc_set_stamp_buffer(buf.data_ptr()) == 0
att = conv.att.detach().reshape(-1).contiguous()
for rep in range(2):
    buf.zero_()
    rc = stamp.isg_gatv2_layer_conv(xp.planes.data_ptr(), xp.inv.data_ptr(), wn.data_ptr(), wn_inv.data_ptr(), cat_b.data_ptr(), ep.data_ptr(),
                                    ep_inv.data_ptr(), we.data_ptr(), we_inv.data_ptr(), att.data_ptr(), conv.bias.data_ptr(),
                                    plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(),
                                    tile_info.data_ptr(), ntiles.data_ptr(), cap, 0, 0, out.data_ptr(), H * C, alpha.data_ptr(),
                                    rowmax.data_ptr(), N, E, H, C, 128, 128, 0.2, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
s = buf.double().cpu()
s = s[s[:, 12] > 0]
names = ["first tile's inputs (once)", "node GEMM: barrier", "chunks: staging barrier", "chunks: k loops",
         "chunks: epilogue + barrier", "next tile: rows -> planes, tables", "softmax + aggregation per node, stores",
         "hand-over barrier", "node GEMM: k loop", "node GEMM: epilogue (LDS writes)", "chunks: wait for the planes, LDS writes", "last logit sums + barrier", "(total)", "(probe) other", "(probe) exposed latency of a chunk request"]
tot = s[:, 12].mean().item()
nwg = s.size(0) // 8
print(f"{T} tiles x {H} heads on {nwg} persistent workgroups of 8 waves; a wave lives {tot:.0f} cycles = {tot * nwg / max(T * H, 1):.0f} per (tile, head)")
for i, n in enumerate(names):
    print(f"  {n:55s} {s[:, i].mean().item():10.0f}  ({100 * s[:, i].mean().item() / tot:5.1f} %)   max {s[:, i].max().item():10.0f}")
