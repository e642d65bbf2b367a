#!/usr/bin/env python3
"""Throughput of the FULL ISubGVQA model (BASELINE configs[2] shape: C = 300, 4 MGAT layers, I-MLE k = 5, question
encoder/decoder included) on GQA-shaped synthetic token batches.   python3 tools/time_full_model.py [graphs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic
from isubgvqa_amd.models import build_model

pol = sys.argv[sys.argv.index('--policy') + 1] if '--policy' in sys.argv else None
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith('--') and sys.argv[i - 1] != '--policy']
graphs = int(args[0]) if len(args) > 0 else 2048
steps = int(args[1]) if len(args) > 1 else 20
if '--policy' in sys.argv:          # isg_linear_h3p's large-result store policy: -1 / 0 / 1 / 2 instead of the measured choice
    ops.H3P_STORE_POLICY = int(pol)
for a_ in sys.argv[1:]:             # --set=NAME=VALUE: any switch of isubgvqa_amd.ops (A/B runs), e.g. --set=GATHER_ADD_PLANES=False
    if a_.startswith('--set='):
        k_, v_ = a_[6:].split('=')
        setattr(ops, k_, eval(v_))
if '--no-h3p' in sys.argv:
    ops.H3P = False        # A/B: the round-3 tile kernel for the K >= 256 Linears
if '--no-chain' in sys.argv:
    ops.H3P_CHAIN = False
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(synthetic.full_model_args(), None).to(dev).eval()
wl = synthetic.make_full_workload(graphs).to(dev)
sg = wl.scene_graphs()


def step():
    return model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True,
                 scene_graphs=sg)[0]


with torch.no_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
print(f"full model, {graphs} graphs (N={wl.x.size(0)}, E={wl.edge_index.size(1)}, T={wl.questions.size(1)}): "
      f"{dt * 1e3:.2f} ms/step = {graphs / dt:,.0f} questions/s; h3p store policy {ops.h3p_store_policy()}")
if '--capture' in sys.argv:       # the same batch through the product path's hipGraph option (ops.StepCapture)
    def cstep():
        return model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg,
                     capture=True)[0]
    with torch.no_grad():
        for _ in range(3):
            cout = cstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            cout = cstep()
        torch.cuda.synchronize()
        dc = (time.perf_counter() - t0) / steps
        model._step_capture.verify()
    print(f"    capture=True: {dc * 1e3:.3f} ms/step = {graphs / dc:,.0f} questions/s; equal to the eager logits: {torch.equal(cout, out)}")

    def lstep():
        return model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg,
                     capture="language")[0]
    with torch.no_grad():
        for _ in range(3):
            lout = lstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lout = lstep()
        torch.cuda.synchronize()
        dl = (time.perf_counter() - t0) / steps
    print(f"    capture='language' (question side replayed, graph side eager): {dl * 1e3:.3f} ms/step = {graphs / dl:,.0f} questions/s; "
          f"equal: {torch.equal(lout, out)}")
