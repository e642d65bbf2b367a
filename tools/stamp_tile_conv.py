#!/usr/bin/env python3
"""Where does a wave of isg_gatv2_tile_conv spend its cycles?  Uses tools/_build/libisg_dt_stamp.so (tools/stamp_dense_tail.py
--build makes it: isg_layer_tile.hip with -DISG_DIAG)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build", "libisg_dt_stamp.so")

import torch

from isubgvqa_amd import _lib, ops, synthetic

stamp = ctypes.CDLL(OUT)
stamp.isg_gatv2_tile_conv.restype, stamp.isg_gatv2_tile_conv.argtypes = _lib.SIGNATURES["isg_gatv2_tile_conv"]
stamp.isg_dt_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(1)
x_lr = torch.randn(N, 2 * H * C, device=dev, generator=g)
x_l, x_r = x_lr[:, :H * C], x_lr[:, H * C:]
w = torch.randn(H * C, C, device=dev, generator=g) * 0.1
att = torch.randn(H * C, device=dev, generator=g)
bias = torch.randn(H * C, device=dev, generator=g)
planes, inv = ops._weight_planes(w, False, "f16x3")
tile_ptr, ntiles, cap, tile_info = plan.tiles(64, 256)
ep, ep_inv = plan.edge_planes(wl.edge_attr)
T = int(ntiles.item())
out = torch.empty(N, H * C, device=dev)
alpha = torch.empty(E, H, device=dev)
rowmax = torch.empty(N, H, device=dev)
nblk = 4096
buf = torch.zeros(nblk * 4, 16, dtype=torch.int64, device=dev)
assert stamp.isg_dt_set_stamp_buffer(buf.data_ptr()) == 0
for rep in range(2):
    buf.zero_()
    rc = stamp.isg_gatv2_tile_conv(x_l.data_ptr(), x_l.stride(0), x_r.data_ptr(), x_r.stride(0), ep.data_ptr(),
                                   ep_inv.data_ptr(), planes.data_ptr(), inv.data_ptr(), att.data_ptr(), bias.data_ptr(),
                                   plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(),
                                   tile_info.data_ptr(), ntiles.data_ptr(), cap, 0, 0,
                                   out.data_ptr(), H * C, alpha.data_ptr(), rowmax.data_ptr(), N, E, H, C, C, 0.2,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
s = buf.double().cpu()
s = s[s[:, 12] > 0]
names = ["first tile's inputs (once)", "hand-over to the next tile (2 barriers, LDS stores)", "chunks: x_r requests + panel staging + barrier",
         "chunks: k loops", "chunks: epilogue + barrier", "-", "C1 softmax weights, alpha + next tile's requests", "C2 aggregation, stores"]
tot = s[:, 12].mean().item()
print(f"{T} tiles x {H} heads on {s.size(0) // 4} persistent workgroups; a wave lives {tot:.0f} cycles = {tot * 4 * H / max(T, 1) / 4:.0f} per (tile, head); "
      f"MFMA floor per wave: {48 * 32} cycles per 64-slot chunk")
for i, n in enumerate(names):
    print(f"  {n:48s} {s[:, i].mean().item():10.0f}  ({100 * s[:, i].mean().item() / tot:5.1f} %)   max {s[:, i].max().item():10.0f}")
