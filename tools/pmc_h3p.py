#!/usr/bin/env python3
"""A few launches of isg_linear_h3p on one shape, for rocprofv3 --pmc passes (L2 hit rate, fetch sizes).
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d DIR -- python3 tools/pmc_h3p.py M N K"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (49152, 1536, 512)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
b = torch.randn(N, device=dev, generator=g)
xp = ops.split_planes32(x)
for _ in range(4):
    ops.linear_h3p(xp, w, b)
torch.cuda.synchronize()
print(f"M={M} N={N} K={K} A_bytes={M * ((K + 31) // 32) * 128} W_bytes={N * ((K + 31) // 32) * 128} D_bytes={M * N * 4}")
