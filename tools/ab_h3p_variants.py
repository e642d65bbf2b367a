#!/usr/bin/env python3
"""Interleaved A/B of isg_linear_h3p builds (tools/_build/libisg_h3p_<name>.so, made with -D switches) against the shipped
library on a few Linear shapes: median us per launch.   python3 tools/ab_h3p_variants.py name [name ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from isubgvqa_amd import _lib, ops

dev = torch.device("cuda:0")
libs = {"shipped": _lib.load()}
PLANES = "--planes" in sys.argv          # time the planes32 result (d_planes / d_inv / d_bound) instead of fp32 rows
for n in [a for a in sys.argv[1:] if not a.startswith("-")]:
    l = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", f"libisg_h3p_{n}.so"))
    l.isg_linear_h3p.restype, l.isg_linear_h3p.argtypes = _lib.SIGNATURES["isg_linear_h3p"]
    libs[n] = l
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K in [(49152, 1536, 512), (49152, 512, 2048), (82189, 2400, 300), (205024, 1200, 300), (16384, 2048, 512)]:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    xp = ops.split_planes32(x)
    wp, winv, bound = ops._h3p_weight(w, b, False)
    out = torch.empty(M, N, device=dev)
    npad = (N + 31) // 32 * 32
    dpl = torch.empty(M * npad * 2, dtype=torch.int16, device=dev)
    dinv = torch.empty(M, device=dev)
    res = {n: [] for n in libs}
    for r in range(9):
        for n, l in libs.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            if PLANES:
                rc = l.isg_linear_h3p(xp.planes.data_ptr(), xp.inv.data_ptr(), wp.data_ptr(), winv.data_ptr(), b.data_ptr(), 0,
                                      dpl.data_ptr(), dinv.data_ptr(), bound.data_ptr(), M, N, K, N, 1, 0, 0, torch.cuda.current_stream().cuda_stream)
            else:
                rc = l.isg_linear_h3p(xp.planes.data_ptr(), xp.inv.data_ptr(), wp.data_ptr(), winv.data_ptr(), b.data_ptr(), out.data_ptr(),
                                      0, 0, 0, M, N, K, N, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
            e.record()
            torch.cuda.synchronize()
            assert rc == 0
            if r >= 2:
                res[n].append(s.elapsed_time(e) * 1e3)
    flop = 6.0 * M * N * K
    print(f"{M} x {N} x {K}: " + "  ".join(f"{n} {sorted(v)[len(v) // 2]:7.1f} us ({flop / sorted(v)[len(v) // 2] / 1e9:.2f} PF/s)" for n, v in res.items()), flush=True)
