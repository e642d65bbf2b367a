#!/usr/bin/env python3
"""Does the speed of isg_linear_h3p under a store cache policy depend on WHERE the result sits?  Times one shape with the result
buffer at several byte offsets inside one allocation, for the shipped library and tools/_build/libisg_h3p_<name>.so variants.
  python3 tools/probe_h3p_store_policy.py M N K name [name ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from isubgvqa_amd import _lib, ops

M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
libs = {"shipped": _lib.load()}
for n in sys.argv[4:]:
    l = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", f"libisg_h3p_{n}.so"))
    l.isg_linear_h3p.restype, l.isg_linear_h3p.argtypes = _lib.SIGNATURES["isg_linear_h3p"]
    libs[n] = l
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
b = torch.randn(N, device=dev, generator=g)
xp = ops.split_planes32(x)
wp, winv, bound = ops._h3p_weight(w, b, False)
pool = torch.empty(M * N + (64 << 20), device=dev)
for off in (0, 1024, 16 << 10, 256 << 10, 1 << 20, 3 << 20, 8 << 20, 33 << 20):
    out = pool[off // 4: off // 4 + M * N].view(M, N)
    line = f"D at +{off:>9d} B (addr mod 2 MiB = {out.data_ptr() % (2 << 20):>8d}):"
    for n, l in libs.items():
        ts = []
        for r in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = l.isg_linear_h3p(xp.planes.data_ptr(), xp.inv.data_ptr(), wp.data_ptr(), winv.data_ptr(), b.data_ptr(), out.data_ptr(),
                                  0, 0, 0, M, N, K, N, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
            e.record()
            torch.cuda.synchronize()
            assert rc == 0
            if r >= 2:
                ts.append(s.elapsed_time(e) * 1e3)
        line += f"  {n} {sorted(ts)[len(ts) // 2]:7.1f} us"
    print(line, flush=True)
