#!/usr/bin/env python3
"""The classifier head of a configs[1] step (cat_mul -> embedding Linear(3C -> 512) + GELU -> logit_fc Linear(512 -> 1842)) at
B = 4096 rows: the shipped dispatch against the planes32 engine forced on both Linears.   python3 tools/time_head.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, C = 4096, 128
embed, glf = torch.randn(B, C, device=dev, generator=g), torch.randn(B, C, device=dev, generator=g)
emb = torch.nn.Sequential(torch.nn.Linear(3 * C, 512), torch.nn.GELU()).to(dev)
fc = torch.nn.Linear(512, 1842).to(dev)


def head():
    feats = ops.mlp(emb, ops.cat_mul(embed, glf), want_rowmax=True)
    return ops.linear(feats, fc.weight, fc.bias)


def bench(fn, n=200):
    with torch.no_grad():
        for _ in range(10):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, out


for label, kw in (("shipped", {}), ("h3p_min_m=2048", {"h3p_min_m": 2048}), ("shipped", {}), ("h3p_min_m=2048", {"h3p_min_m": 2048})):
    with ops.configured(**kw):
        us, out = bench(head)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with ops.configured(**kw), torch.no_grad():
        ev[0].record()
        for _ in range(50):
            head()
        ev[1].record()
    torch.cuda.synchronize()
    print(f"{label:16s}: {us:7.1f} us per head (wall, back to back), {ev[0].elapsed_time(ev[1]) / 50 * 1e3:7.1f} us on the stream; logits[0,:3] = {out[0, :3].tolist()}")
