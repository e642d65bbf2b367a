#!/usr/bin/env python3
"""Per-k-tile cost of isg_linear_bf16x6: time vs K at fixed M, N (slope = main loop, intercept = prologue + epilogue)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

dev = torch.device("cuda:0")
M, N = 65536, 256
g = torch.Generator(device=dev).manual_seed(0)
for gelu in (False, True):
    for K in (128, 256, 512, 1024, 2048):
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        b = torch.randn(N, device=dev, generator=g)
        ts = []
        for r in range(8):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.linear(x, w, b, gelu=gelu)
            e.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(s.elapsed_time(e) * 1e3)
        t = sorted(ts)[len(ts) // 2]
        print(f"gelu={int(gelu)} K={K:5d}  {t:8.1f} us   {2.0 * M * K * N / t / 1e6:7.1f} TF")
