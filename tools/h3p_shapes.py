#!/usr/bin/env python3
"""Every isg_linear_h3p launch of one full-model step (BASELINE configs[2] stand-in) by shape: launches, HIP-event time, PF/s of
fp16 products (3 per multiply-add, K padded as the kernel walks it).   python3 tools/h3p_shapes.py [graphs] [steps]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic
from isubgvqa_amd.models import build_model

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(synthetic.full_model_args(), None).to(dev).eval()
wl = synthetic.make_full_workload(graphs).to(dev)
sg = wl.scene_graphs()
step = lambda: model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)[0]
with torch.no_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ops.H3P_TIMER = ops.KernelTimer()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
t = ops.H3P_TIMER
ops.H3P_TIMER = None
acc = collections.OrderedDict()
for ms, m in zip(t.durations_ms(), t.meta):
    key = (m["M"], m["N"], m["K"], m["planes_out"])
    a = acc.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += ms
tot = sum(a[1] for a in acc.values()) / steps
print(f"{graphs} graphs: {sum(a[0] for a in acc.values()) // steps} launches, {tot:.3f} ms per step on the engine")
print(f"{'M':>8s} {'N':>6s} {'K':>6s} planes  launches/step   us/launch   ms/step    PF/s")
for (M, N, K, po), (n, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    kp = (K + 31) // 32 * 32
    us = ms / n * 1e3
    print(f"{M:8d} {N:6d} {K:6d} {str(po):6s} {n / steps:10.1f} {us:12.1f} {ms / steps:9.3f} {6.0 * M * N * kp / (us * 1e-6) / 1e15:8.3f}")
