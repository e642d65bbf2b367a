#!/usr/bin/env python3
"""Where does a wave of isg_mgat_dense_tail spend its cycles?  `--build` (in the build container) makes
tools/_build/libisg_dt_stamp.so from isg_layer_tile.hip with -DISG_DIAG; the run launches it once at the BASELINE configs[1]
shapes and prints the mean core-clock cycles per phase over all waves."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "libisg_dt_stamp.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DISG_DIAG",
                           *[a for a in sys.argv[1:] if a.startswith("-D")],
                           os.path.join(CSRC, "isg_layer_tile.hip"), os.path.join(CSRC, "isg_layer_conv.hip"), os.path.join(CSRC, "isg_graph.hip"),
                           "-o", OUT])
    print("built", OUT)
    sys.exit(0)

import torch

from isubgvqa_amd import _lib, ops, synthetic

stamp = ctypes.CDLL(OUT)
stamp.isg_mgat_dense_tail.restype, stamp.isg_mgat_dense_tail.argtypes = _lib.SIGNATURES["isg_mgat_dense_tail"]
stamp.isg_dt_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
net = synthetic.build_answer_model(cfg).to(dev).eval()
m = net.gat_seq
N, H, C = wl.x.size(0), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(1)
conv_out = torch.randn(N, H * C, device=dev, generator=g)
rm = conv_out.view(N, H, C).abs().amax(dim=2).contiguous()
h = torch.randn(N, C, device=dev, generator=g)
ins, ins_next = wl.instr[0].contiguous(), wl.instr[1].contiguous()
bn = m.bns[0]
tile_ptr, ntiles, cap, tile_info = plan.tiles(64)
T = int(ntiles.item())
l0, l2 = m.x_proj[0][0], m.x_proj[0][2]
p1, inv1 = ops._weight_planes(l0.weight, True, "f16x3")
p2, inv2 = ops._weight_planes(l2.weight, True, "f16x3")
yb = torch.stack([l0.weight.detach().abs().sum(dim=1).max(), l0.bias.detach().abs().max()]).float().contiguous()
h_out = torch.empty_like(h)      # the next layer's input as planes, as the model runs it
xp, xinv = torch.empty(N, 2, 128, dtype=torch.int16, device=dev), torch.empty(N, device=dev)
buf = torch.zeros(cap * 4, 16, dtype=torch.int64, device=dev)
assert stamp.isg_dt_set_stamp_buffer(buf.data_ptr()) == 0
for rep in range(2):
    buf.zero_()
    rc = stamp.isg_mgat_dense_tail(conv_out.data_ptr(), conv_out.stride(0), rm.data_ptr(), rm.size(1), rm.stride(0), p1.data_ptr(),
                                   inv1.data_ptr(), l0.bias.data_ptr(), yb.data_ptr(), p2.data_ptr(), inv2.data_ptr(), l2.bias.data_ptr(),
                                   ins.data_ptr(), h.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.mean_scale.data_ptr(),
                                   float(bn.eps), 0, ins_next.data_ptr(), h_out.data_ptr(), 0, xp.data_ptr(), xinv.data_ptr(), plan.ptr.data_ptr(),
                                   wl.batch.data_ptr(), tile_ptr.data_ptr(), tile_info.data_ptr(), ntiles.data_ptr(), cap, N, 512, 256, 128,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
s = buf[:T * 4].double().cpu()
names = ["tile header + row scales", "chunk 0 staged", "chunk 0 computed + 1 staged", "chunk 1 computed + 2 staged",
         "chunk 2 computed + 3 staged", "chunk 3 computed", "epilogue 1 (GELU, maxima, planes)", "GEMM2", "epilogue 2",
         "tail A (logits)", "tail B (softmax)", "tail C (norm, residual, gate)", "whole kernel"]
tot = s[:, 12].mean().item()
print(f"{T} tiles, mean rows {s[:, 13].mean().item():.1f}, mean graphs {s[:, 14].mean().item():.2f}; a wave lives {tot:.0f} cycles "
      f"(100 MHz counter? see below); MFMA issue floor: 480 x 32 = 15360 core cycles")
for i, n in enumerate(names):
    print(f"  {n:38s} {s[:, i].mean().item():10.0f}  ({100 * s[:, i].mean().item() / tot:5.1f} %)   max {s[:, i].max().item():10.0f}")
