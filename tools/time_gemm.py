#!/usr/bin/env python3
"""A/B timing of the dense projections of one MGAT layer at BASELINE configs[1] sizes: hipBLASLt fp32 through torch
vs isg_linear_bf16x6 (interleaved rounds, HIP events).   python3 tools/time_gemm.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
N_, E_ = 82286, 205024
shapes = [("lin_l/lin_r  [N,128]x[512,128]", N_, 128, 512, False), ("lin_edge     [E,128]x[512,128]", E_, 128, 512, False),
          ("x_proj.0+gelu[N,512]x[256,512]", N_, 512, 256, True), ("x_proj.2+gelu[N,256]x[128,256]", N_, 256, 128, True),
          ("node_nn+gelu [N,128]x[128,128]", N_, 128, 128, True), ("classifier   [4096,512]x[1842,512]", 4096, 512, 1842, False)]
g = torch.Generator(device=dev).manual_seed(0)
for name, M, K, N, gelu in shapes:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    res = {"torch": [], "bf16x6": []}
    for r in range(rounds + 2):
        for be in ("torch", "bf16x6"):
            ops.GEMM_BACKEND = be
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = ops.linear(x, w, b, gelu=gelu)
            e.record()
            torch.cuda.synchronize()
            if r >= 2:
                res[be].append(s.elapsed_time(e) * 1e3)
    fl = 2.0 * M * K * N
    t, k = sorted(res["torch"])[len(res["torch"]) // 2], sorted(res["bf16x6"])[len(res["bf16x6"]) // 2]
    print(f"{name:38s} torch {t:7.1f} us ({fl / t / 1e6:6.1f} TF)   bf16x6 {k:7.1f} us ({fl / k / 1e6:6.1f} TF)   x{t / k:.2f}")
