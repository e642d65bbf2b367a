#!/usr/bin/env python3
"""Host time per call of the hottest operators at a tiny size (nothing waited for: issue cost only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isubgvqa_amd import ops, _lib
dev = torch.device("cuda:0")
x = torch.randn(12, 512, device=dev); w = torch.randn(512, 512, device=dev); b = torch.randn(512, device=dev)
lin = torch.nn.Linear(512, 512).to(dev)
seq = torch.nn.Sequential(torch.nn.Linear(512, 512), torch.nn.GELU()).to(dev)
r = torch.randn(12, 512, device=dev)
norm = torch.nn.LayerNorm(512).to(dev)
lib = _lib.load()
out = torch.empty(12, 512, device=dev)
def t(name, f, n=3000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    print(f"{name:50s} {dt * 1e6:7.2f} us/call", flush=True)
with torch.no_grad():
    st = ops._stream()
    t("raw ctypes isg_linear_skinny", lambda: lib.isg_linear_skinny(x.data_ptr(), 512, w.data_ptr(), 512, b.data_ptr(), out.data_ptr(), 512, 12, 512, 512, 0, st))
    t("torch.empty(12, 512)", lambda: torch.empty(12, 512, dtype=torch.float32, device=dev))
    t("ops._stream()", lambda: ops._stream())
    t("ops.linear_skinny(x, w, b)", lambda: ops.linear_skinny(x, w, b))
    t("ops.linear(x, w, b)", lambda: ops.linear(x, w, b))
    t("ops.linear(x, lin.weight, lin.bias)", lambda: ops.linear(x, lin.weight, lin.bias))
    t("ops.mlp(seq, x)", lambda: ops.mlp(seq, x))
    t("ops.add_layernorm(x, r, norm)", lambda: ops.add_layernorm(x, r, norm))
    t("torch F.linear (hipBLASLt)", lambda: torch.nn.functional.linear(x, w, b))
