#!/usr/bin/env python3
"""dW = g^T x: isg_linear_wgrad (fp32 MFMA, split over rows) vs torch (hipBLASLt fp32), interleaved, HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

dev = torch.device("cuda:0")
for name, M, N, K in [("lin_edge", 205024, 512, 128), ("lin_l|lin_r", 82286, 1024, 128), ("x_proj.0", 82286, 256, 512),
                      ("x_proj.2", 82286, 128, 256), ("node_nn", 82286, 128, 128), ("logit_fc", 4096, 1842, 512)]:
    g, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    res = {"torch": [], "isg": []}
    for r in range(10):
        for k in res:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = g.t() @ x if k == "torch" else ops.linear_wgrad(g, x)
            e.record()
            torch.cuda.synchronize()
            if r >= 2:
                res[k].append(s.elapsed_time(e) * 1e3)
    t, i = (sorted(v)[len(v) // 2] for v in (res["torch"], res["isg"]))
    fl = 2.0 * M * N * K
    print(f"{name:12s} [{M},{N}]^T x [{M},{K}]: torch {t:7.1f} us ({fl / t / 1e6:5.1f} TF)  isg {i:7.1f} us ({fl / i / 1e6:5.1f} TF)  x{t / i:.2f}")
