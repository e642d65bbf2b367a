#!/usr/bin/env python3
"""isg_linear_h3p (planes32 engine) against the shipped tile kernel: error vs an fp64 product and time per launch,
interleaved on one device, on the full model's Linear shapes.   python3 tools/time_h3p.py [--quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [  # M, N, K, act, what
    (49152, 1536, 512, None, "encoder in_proj"),
    (49152, 512, 512, None, "encoder out_proj"),
    (49152, 2048, 512, "relu", "encoder linear1"),
    (49152, 512, 2048, None, "encoder linear2"),
    (49152, 1024, 512, None, "decoder cross kv"),
    (16384, 2048, 512, "relu", "decoder linear1"),
    (82189, 2400, 300, None, "lin_l|lin_r C=300"),
    (82189, 600, 1200, "gelu", "x_proj.0 C=300"),
    (82189, 300, 600, "gelu", "x_proj.2 C=300"),
    (205024, 1200, 300, None, "lin_edge C=300"),
]
if "--quick" in sys.argv:
    SHAPES = SHAPES[:4]
if "--n300" in sys.argv:       # the full model's Linears with 300 result columns (256-wide column tiles: the second 44 columns full)
    SHAPES = [(204753, 300, 300, None, "edge encoder"), (82189, 300, 600, "gelu", "x_proj.2"), (82189, 300, 300, None, "node 300x300"),
              (82189, 320, 300, None, "(N = 320)"), (82189, 256, 300, None, "(N = 256)"), (82189, 512, 300, None, "(N = 512)")]
if "--small" in sys.argv:       # the configs[1] step's classifier side (below ops.H3P_MIN_M rows)
    SHAPES = [(4096, 256, 256, "gelu", "embedding"), (4096, 1840, 256, None, "logit_fc (1842)"), (4096, 1844, 256, None, "logit_fc (1842)"),
              (8192, 1844, 256, None, "logit_fc x2"), (4096, 1844, 512, None, "logit_fc K=512")]


def timed(fn, reps=7):
    ts = []
    for r in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


for M, N, K, act, what in SHAPES:
    x = torch.randn(M, K, device=dev, generator=g) * torch.rand(M, 1, device=dev, generator=g).mul(4).exp()
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    kw = dict(gelu=act == "gelu", relu=act == "relu")
    rows = slice(0, 4096)
    ref = x[rows].double() @ w.double().t() + b.double()
    base = x[rows] @ w.t() + b
    f = {"gelu": torch.nn.functional.gelu, "relu": torch.relu}.get(act, lambda t: t)
    ref, base = f(ref), f(base)
    e32 = (base.double() - ref).abs().max().item()
    xp = ops.split_planes32(x)
    new = ops.linear_h3p(xp, w, b, **kw)
    ops.H3P = False
    old = ops.linear(x, w, b, **kw)
    ops.H3P = True
    en = (new[rows].double() - ref).abs().max().item()
    eo = (old[rows].double() - ref).abs().max().item()
    line = f"{what:20s} {M:6d} x {N:4d} x {K:4d}: err/fp32 new {en / e32:5.2f} old {eo / e32:5.2f}"
    if N % 32 == 0:
        pl = ops.linear_h3p(xp, w, b, planes_out=True, **kw)
        ep = (ops.planes32_to_rows(pl)[rows].double() - ref).abs().max().item()
        line += f" planes-out {ep / e32:5.2f}"
    flop = 2.0 * M * N * K * 3
    t_split = timed(lambda: (setattr(x, "_isg_planes32", None), ops.split_planes32(x)))
    t_new = timed(lambda: ops.linear_h3p(xp, w, b, **kw))
    ops.H3P = False
    t_old = timed(lambda: ops.linear(x, w, b, **kw))
    ops.H3P = True
    line += (f" | new {t_new:7.1f} us = {flop / t_new / 1e9:6.3f} PF/s  (+split {t_split:6.1f})  old {t_old:7.1f} us = "
             f"{flop / t_old / 1e9:6.3f} PF/s")
    if N % 32 == 0:
        t_pl = timed(lambda: ops.linear_h3p(xp, w, b, planes_out=True, **kw))
        line += f"  planes-out {t_pl:7.1f} us"
    print(line, flush=True)
    del x, w, b, new, old, xp
    torch.cuda.empty_cache()
