#!/usr/bin/env python3
"""isg_linear_f16x3_tile alone at the shapes that carry the full model and x_proj: (rows, N, K), row maxima handed in.
ISG_F16X3_MFMA=32 selects the 32x32x16 MFMA arm (default: 16x16x32); run both in one gpurun call for a same-box A/B.
    python3 tools/time_f16x3_tile.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import _lib, ops

if os.environ.get("ISG_TOOL_LIB"):        # A/B against another build of the library (same box, separate processes)
    _lib.LIB_PATH = os.path.abspath(os.environ["ISG_TOOL_LIB"])

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("text in_proj", 49152, 1536, 512), ("text out_proj / q", 49152, 512, 512), ("text linear1 (+relu)", 49152, 2048, 512),
          ("x_proj.0 cfg2", 82286, 512, 512), ("x_proj.2 cfg2", 82286, 128, 256), ("lin_l|r C=300", 82189, 2400, 300)]
flush = torch.empty(1 << 27, device=dev)
print(f"MFMA arm: {os.environ.get('ISG_F16X3_MFMA', '16 (default)')}")
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    ops.attach_row_maxima(x, x.abs().amax(1, keepdim=True).contiguous())
    ts = []
    for r in range(rounds + 2):
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = ops.linear(x, w, b, relu="relu" in name)
        e.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(s.elapsed_time(e) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f"{name:22s} [{M} x {N} x {K}]  median {t:7.1f} us  min {min(ts):7.1f} us   {6.0 * M * N * K / t / 1e6:7.1f} TFLOP/s of fp16 products")
