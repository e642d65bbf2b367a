#!/usr/bin/env python3
"""Where does isg_gatv2_edge_logits spend its time?  `--build` (in the build container) makes tools/_build/libisg_el_abl.so
from isg_mp_logits.hip with -DISG_EL_ABLATION; the run times compile-time ablated variants (ISG_EL_DBG) interleaved in one
process with HIP events at the BASELINE configs[1] topology.
  DBG bits: 1 no x_l / x_r row gathers, 2 no MFMAs.  (The first version of the kernel also had bits for the epilogue, the W
  loads and the panel staging; its numbers are in profiles/r02_o_el_ablation.txt.)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "libisg_el_abl.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                           "-DISG_EL_ABLATION", os.path.join(CSRC, "isg_mp_logits.hip"), os.path.join(CSRC, "isg_graph.hip"), "-o", OUT])
    print("built", OUT)
    sys.exit(0)

import torch

from isubgvqa_amd import ops, synthetic

lib = ctypes.CDLL(OUT)
c = ctypes
lib.isg_gatv2_edge_logits.argtypes = [c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int64, c.c_void_p,
                                      c.c_int32, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                                      c.c_void_p, c.c_int64, c.c_int32, c.c_int32, c.c_int32, c.c_float, c.c_void_p]
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
K = wl.edge_attr.size(1)
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
plan.require_csr()
g = torch.Generator(device=dev).manual_seed(0)
x_lr = torch.randn(N, 2 * H * C, device=dev, generator=g)
ea = wl.edge_attr.float().contiguous()
w = torch.randn(H * C, K, device=dev, generator=g) / K ** 0.5
att = torch.randn(H * C, device=dev, generator=g)
planes, inv = ops._weight_planes(w, True, "f16x3")
lg = torch.empty(E, H, device=dev)
flush = torch.empty(1 << 27, device=dev)
st = torch.cuda.current_stream().cuda_stream
variants = [(0, "full"), (1, "no row gathers"), (2, "no MFMA"), (3, "no gathers, no MFMA")]
res = {v: [] for v, _ in variants}
for r in range(10):
    for v, _ in variants:
        os.environ["ISG_EL_DBG"] = str(v)
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = lib.isg_gatv2_edge_logits(ea.data_ptr(), K, planes.data_ptr(), inv.data_ptr(), x_lr.data_ptr(), 2 * H * C, 0,
                                       x_lr.data_ptr() + 4 * H * C, 2 * H * C, 0, att.data_ptr(), plan.eid.data_ptr(),
                                       plan.src.data_ptr(), plan.dst.data_ptr(), None, None, lg.data_ptr(), E, H, C, K, 0.2, st)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0
        if r >= 2:
            res[v].append(s.elapsed_time(e) * 1e3)
print(f"isg_gatv2_edge_logits N={N} E={E} H={H} C={C} K={K}")
for v, label in variants:
    t = sorted(res[v])[len(res[v]) // 2]
    print(f"   DBG={v:2d} {label:36s} {t:8.1f} us", flush=True)
