#!/usr/bin/env python3
"""A/B timing of the dense projections of one MGAT layer at BASELINE configs[1] sizes (and the reference-width C = 300
shapes): hipBLASLt fp32 through torch vs the tile kernel (isg_linear_bf16x6) vs the row-panel kernel (isg_linear_panel);
interleaved rounds in one process, HIP events.   python3 tools/time_gemm.py [rounds] [c300]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
N_, E_ = 82286, 205024
shapes = [("lin_l|lin_r  [N,128]x[1024,128]", N_, 128, 1024, False), ("lin_edge     [E,128]x[512,128]", E_, 128, 512, False),
          ("x_proj.0+gelu[N,512]x[256,512]", N_, 512, 256, True), ("x_proj.2+gelu[N,256]x[128,256]", N_, 256, 128, True),
          ("node_nn+gelu [N,128]x[128,128]", N_, 128, 128, True), ("classifier   [4096,512]x[1842,512]", 4096, 512, 1842, False)]
if "c300" in sys.argv:
    shapes = [("lin_l|lin_r  [N,300]x[2400,300]", N_, 300, 2400, False), ("lin_edge     [E,300]x[1200,300]", E_, 300, 1200, False),
              ("x_proj.0+gelu[N,1200]x[600,1200]", N_, 1200, 600, True), ("x_proj.2+gelu[N,600]x[300,600]", N_, 600, 300, True),
              ("text in_proj [49152,512]x[1536,512]", 49152, 512, 1536, False), ("text ffn1+relu~[49152,512]x[2048,512]", 49152, 512, 2048, False),
              ("text ffn2    [49152,2048]x[512,2048]", 49152, 2048, 512, False)]
g = torch.Generator(device=dev).manual_seed(0)
arms = ("torch", "tile", "panel", "f16x3")
for name, M, K, N, gelu in shapes:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    res = {a: [] for a in arms}
    for r in range(rounds + 2):
        for arm in arms:
            ops.GEMM_BACKEND = "torch" if arm == "torch" else "bf16x6"
            ops.GEMM_KERNEL = {"tile": "tile", "panel": "panel", "f16x3": "panel"}.get(arm, ops.GEMM_KERNEL)
            ops.GEMM_F16X3 = arm == "f16x3"
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = ops.linear(x, w, b, gelu=gelu)
            e.record()
            torch.cuda.synchronize()
            if r >= 2:
                res[arm].append(s.elapsed_time(e) * 1e3)
    fl = 2.0 * M * K * N
    med = {a: sorted(v)[len(v) // 2] for a, v in res.items()}
    mn = {a: min(v) for a, v in res.items()}
    hbm = 4.0 * (M * K + M * N)
    print(f"{name:38s} " + "  ".join(f"{a} {med[a]:7.1f} us (min {mn[a]:6.1f}; {fl / med[a] / 1e6:6.1f} TF, {hbm / med[a] / 1e3:5.0f} GB/s)" for a in arms)
          + f"   f16x3/tile x{med['tile'] / med['f16x3']:.2f}", flush=True)
