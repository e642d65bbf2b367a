#!/usr/bin/env python3
"""Same-process A/B of the configs[1] step (bench.py's workload) under the library's switches: boxes differ by ~10 %, so
variants are interleaved round by round on ONE device and the median per variant is reported.
  python tools/ab_step.py [rounds] [steps_per_round]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
model = synthetic.build_answer_model(cfg).to(dev).eval()
variants = [
    ("r01: tile GEMM everywhere", dict(GEMM_KERNEL="tile", LINEAR_MULTI=False, GEMM_F16X3=False, F16X3_TILE=False, FUSE_LOGITS=False)),
    ("+ row-panel GEMM for the K=128 projections", dict(GEMM_KERNEL="auto", LINEAR_MULTI=False, GEMM_F16X3=False, F16X3_TILE=False, FUSE_LOGITS=False)),
    ("+ the layers' lin_edge as one multi-output launch", dict(GEMM_KERNEL="auto", LINEAR_MULTI=True, GEMM_F16X3=False, F16X3_TILE=False, FUSE_LOGITS=False)),
    ("+ fp16 three-product form for those projections", dict(GEMM_KERNEL="auto", LINEAR_MULTI=True, GEMM_F16X3=True, F16X3_TILE=False, FUSE_LOGITS=False)),
    ("+ the same for x_proj (row maxima from the producer kernels)", dict(GEMM_KERNEL="auto", LINEAR_MULTI=True, GEMM_F16X3=True, F16X3_TILE=True, FUSE_LOGITS=False)),
    ("+ lin_edge folded into the logits (no e_proj in memory)", dict(GEMM_KERNEL="auto", LINEAR_MULTI=True, GEMM_F16X3=True, F16X3_TILE=True, FUSE_LOGITS=True)),
]
if os.environ.get("AB_R03"):         # round 3's switches on top of round 2's default path
    base = dict(GEMM_KERNEL="auto", LINEAR_MULTI=True, GEMM_F16X3=True, F16X3_TILE=True, FUSE_LOGITS=True)
    variants = [
        ("r02 default (edge-logits pair, un-fused tail, 14-launch plan)", {**base, "FUSE_TILE_CONV": False, "FUSE_DENSE_TAIL": False, "PLAN_FUSED": False}),
        ("+ fused dense tail", {**base, "FUSE_TILE_CONV": False, "FUSE_DENSE_TAIL": True, "PLAN_FUSED": False}),
        ("+ tile conv", {**base, "FUSE_TILE_CONV": True, "FUSE_DENSE_TAIL": True, "PLAN_FUSED": False}),
        ("+ 6-launch plan build", {**base, "FUSE_TILE_CONV": True, "FUSE_DENSE_TAIL": True, "PLAN_FUSED": True, "FUSE_LAYER_CONV": False}),
        ("+ lin_l | lin_r inside the conv kernel (layer conv)", {**base, "FUSE_TILE_CONV": True, "FUSE_DENSE_TAIL": True, "PLAN_FUSED": True, "FUSE_LAYER_CONV": True}),
        ("+ read-out on tiles (r03 default)", {**base, "FUSE_TILE_CONV": True, "FUSE_DENSE_TAIL": True, "PLAN_FUSED": True, "FUSE_LAYER_CONV": True, "FUSE_READOUT": True}),
    ]
    for name, sw in variants[:3]:
        sw["FUSE_LAYER_CONV"] = False
    for name, sw in variants[:5]:
        sw["FUSE_READOUT"] = False
if os.environ.get("AB_PANEL_N"):     # sweep the narrowest Linear the panel kernels take, on the default path
    base = dict(variants[-2][1])
    variants = [(f"default path, PANEL_MIN_N = {n}", {**base, "PANEL_MIN_N": n}) for n in (256, 128, 64, 256)]
if os.environ.get("AB_LAST"):       # only the last N variants (short runs)
    variants = variants[-int(os.environ["AB_LAST"]):]
res = {name: [] for name, _ in variants}
with torch.no_grad():
    for r in range(rounds + 1):
        for name, sw in variants:
            for k, v in sw.items():
                setattr(ops, k, v)
            for i in range(3):
                model(wl, seed=i, use_hints=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                model(wl, seed=100 + i, use_hints=True)
            torch.cuda.synchronize()
            if r > 0:
                res[name].append((time.perf_counter() - t0) / steps * 1e3)
for k, v in dict(GEMM_KERNEL="auto", LINEAR_MULTI=True, GEMM_F16X3=True, F16X3_TILE=True, FUSE_LOGITS=True,
                 FUSE_TILE_CONV=True, FUSE_DENSE_TAIL=True, PLAN_FUSED=True, FUSE_LAYER_CONV=True, FUSE_READOUT=True).items():
    setattr(ops, k, v)
for name, _ in variants:
    t = sorted(res[name])
    print(f"{name:55s} median {t[len(t) // 2]:.3f} ms/step (min {t[0]:.3f})  = {cfg.num_graphs / t[len(t) // 2] * 1e3:,.0f} q/s")
