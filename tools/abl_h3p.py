#!/usr/bin/env python3
"""Where a tile's time goes in isg_linear_h3p: ISG_P3_ABL=1 drops the result stores, =2 runs two k-tiles only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isubgvqa_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, reps=9):
    ts = []
    for r in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]
ABL = sys.argv[1:] or ['0', '1', '4', '8', '12', '13']
for M, N, K in [(49152, 1536, 512), (65536, 1024, 512), (65536, 1024, 1024), (65536, 1024, 2048), (65536, 256, 512), (16384, 1024, 512)]:
    x = torch.randn(M, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    xp = ops.split_planes32(x)
    out = []
    for abl in ABL:
        os.environ["ISG_P3_ABL"] = abl
        out.append(timed(lambda: ops.linear_h3p(xp, w, b)))
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{M} x {N} x {K}: " + "  ".join(f"abl {a}: {t:7.1f}" for a, t in zip(ABL, out)), flush=True)
