#!/usr/bin/env python3
"""x_proj of one MGAT layer at configs[1] (Linear 512 -> 256 + GELU, Linear 256 -> 128 + GELU over 82 286 rows): the bf16
six-product tile kernel against the fp16 three-product tile kernel with row maxima from the producer.  HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M = 82286
x = torch.randn(M, 512, device=dev, generator=g)
w0, b0 = torch.randn(256, 512, device=dev, generator=g) / 512 ** 0.5, torch.randn(256, device=dev, generator=g)
w1, b1 = torch.randn(128, 256, device=dev, generator=g) / 16, torch.randn(128, device=dev, generator=g)
rm = x.view(M, 4, 128).abs().amax(2).contiguous()
res = {}
for r in range(10):
    for name, f16, gelu in (("bf16x6 +gelu", False, True), ("bf16x6 plain", False, False),
                            ("f16x3 tile +gelu", True, True), ("f16x3 tile plain", True, False)):
        ops.F16X3_TILE = f16
        xx = x.view(M, 512)
        if f16:
            xx = x.clone() if False else x
            ops.attach_row_maxima(xx, rm)
        elif hasattr(x, "_isg_rowmax"):
            del x._isg_rowmax
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        h = ops.linear(xx, w0, b0, gelu=gelu, want_rowmax=f16)
        ev[1].record()
        y = ops.linear(h, w1, b1, gelu=gelu)
        ev[2].record()
        torch.cuda.synchronize()
        if r >= 2:
            res.setdefault(name, []).append((ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3))
for name, v in res.items():
    a = sorted(t[0] for t in v)[len(v) // 2]
    b = sorted(t[1] for t in v)[len(v) // 2]
    print(f"{name:22s} x_proj.0 {a:6.1f} us   x_proj.2 {b:6.1f} us")
