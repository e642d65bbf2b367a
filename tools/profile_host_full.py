#!/usr/bin/env python3
"""Where the HOST spends a forward of the FULL model at a small batch (the regime where the GPU is no longer the limit: DESIGN 17.9).
  python3 tools/profile_host_full.py [graphs]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic
from isubgvqa_amd.models import build_model

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(synthetic.full_model_args(), None).to(dev).eval()
wl = synthetic.make_full_workload(graphs).to(dev)
sg = wl.scene_graphs()
step = lambda: model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)[0]
with torch.no_grad():
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    t_issue = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    print(f"full model, {graphs} graphs: host issue {t_issue * 1e3:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20):
        step()
        if i % 4 == 3:
            torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(60)
