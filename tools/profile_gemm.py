#!/usr/bin/env python3
"""Run isg_linear_bf16x6 a few times at one shape (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 1024, 256)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
b = torch.randn(N, device=dev, generator=g)
for _ in range(4):
    ops.linear(x, w, b)
torch.cuda.synchronize()
