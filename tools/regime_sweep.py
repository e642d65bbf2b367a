#!/usr/bin/env python3
"""A sweep over unusual shapes of the FULL model: long questions, big graphs, tiny and huge batches -- ms per forward and a finiteness
check; what a perf cliff or an unsupported shape would show up in.   python3 tools/regime_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic
from isubgvqa_amd.models import build_model

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(synthetic.full_model_args(), None).to(dev).eval()
cases = [("1 question, 30 tokens", dict(num_graphs=1, tokens=30)), ("8 questions, 20 tokens", dict(num_graphs=8, tokens=20)),
         ("64 questions, 40 tokens", dict(num_graphs=64, tokens=40)), ("4 graphs of 150 nodes", dict(num_graphs=4, sizes=(150,) * 4)),
         ("256 graphs, 16 of 100-190 nodes", dict(num_graphs=256, sizes=tuple([20] * 240 + [100 + 6 * i for i in range(16)]))),
         ("2048 graphs, 77 tokens", dict(num_graphs=2048, tokens=77)), ("16384 graphs, 12 tokens", dict(num_graphs=16384))]
for name, kw in cases:
    wl = synthetic.make_full_workload(**kw).to(dev)
    sg = wl.scene_graphs()
    f = lambda: model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)
    with torch.no_grad():
        ops.reset_counters()
        for _ in range(3):
            out = f()
        torch.cuda.synchronize()
        c = ops.counters()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            out = f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    ok = bool(torch.isfinite(out[0]).all())
    print(f"{name:36s} N={wl.x.size(0):7d} E={wl.edge_index.size(1):8d} T={wl.questions.size(1):3d}: {dt * 1e3:8.2f} ms  "
          f"({wl.questions.size(0) / dt:10,.0f} q/s)  finite={ok}  torch fallbacks: linear {c['torch_linear'] // 3} attention {c['torch_attention'] // 3} "
          f"layer_norm {c['torch_layer_norm'] // 3}", flush=True)
ops.check_plans()
