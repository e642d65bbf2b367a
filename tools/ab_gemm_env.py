#!/usr/bin/env python3
"""Same-process A/B of isg_linear_bf16x6 under an environment switch read at launch time (e.g. ISG_GEMM_NO_XCD),
interleaved launches, HIP events.   python3 tools/ab_gemm_env.py ISG_GEMM_NO_XCD"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

var = sys.argv[1]
dev = torch.device("cuda:0")
shapes = [(82286, 128, 1024, False), (205024, 128, 512, False), (82286, 512, 256, True), (82286, 256, 128, True),
          (82286, 128, 128, True), (4096, 512, 1842, False)]
for M, K, N, gelu in shapes:
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    res = {"off": [], "on": []}
    outs = {}
    for r in range(14):
        for mode in ("off", "on"):
            if mode == "off":
                os.environ[var] = "1"
            else:
                os.environ.pop(var, None)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            outs[mode] = ops.linear(x, w, b, gelu=gelu)
            e.record()
            torch.cuda.synchronize()
            if r >= 2:
                res[mode].append(s.elapsed_time(e) * 1e3)
    off, on = (sorted(v)[len(v) // 2] for v in (res["off"], res["on"]))
    print(f"M={M} K={K} N={N}: with {var} {off:7.1f} us   default {on:7.1f} us   x{off / on:.3f}   identical={torch.equal(outs['off'], outs['on'])}")
