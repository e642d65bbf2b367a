#!/usr/bin/env python3
"""Bit-compare the per-graph kernels (layer tail, scatter attention, GraphNorm, attention pooling) of a variant build of
libisg_hip.so with the shipped one on random skewed graphs.   python3 tools/cmp_per_graph_variant.py tools/_build/libisg_variant.so"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from isubgvqa_amd import _lib, ops

dev = torch.device("cuda:0")
shipped = _lib.load()
variant = ctypes.CDLL(os.path.join(ROOT, sys.argv[1]))
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(variant, name)
    fn.restype, fn.argtypes = res, args
gen = torch.Generator().manual_seed(5)
for C in (128, 300, 20):
    sizes = torch.randint(1, 200, (300,), generator=gen)
    sizes[7] = 260 if C == 20 else 200
    batch = torch.repeat_interleave(torch.arange(sizes.numel()), sizes).to(dev)
    N, B = batch.numel(), sizes.numel()
    plan = ops.GraphPlan.build(batch, None, num_graphs=B)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)
    ins, c, h, w, b, ms, q = r(B, C), r(N, C), r(N, C), r(C), r(C), r(C), r(B, C)
    nm = (torch.rand(N, 1, generator=gen) > 0.5).float().to(dev)
    res = {}
    for name, lib in (("shipped", shipped), ("variant", variant)):
        _lib._lib = lib
        res[name] = [ops.scatter_attention(ins, c, plan), ops.graph_norm(c, plan, w, b, ms),
                     ops.mgat_layer_tail(ins, c, h, plan, w, b, ms), ops.mgat_layer_tail(ins, c, h, plan, w, b, ms, node_mask=nm),
                     *ops.global_attn_pool(c, q, plan, None), *ops.global_attn_pool(c, q, plan, nm)]
        torch.cuda.synchronize()
    _lib._lib = shipped
    names = ["scatter_attention", "graph_norm", "layer_tail", "layer_tail masked", "pool out", "pool gate", "pool out masked", "pool gate masked"]
    print(C, {n: ("same" if torch.equal(a, v) else f"{(a - v).abs().max().item():.2e}") for n, a, v in zip(names, res["shipped"], res["variant"])})
