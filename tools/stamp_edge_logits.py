#!/usr/bin/env python3
"""Where do the waves of isg_gatv2_edge_logits spend their cycles?  `--build` (build container) makes
tools/_build/libisg_el_stamp.so from isg_mp_logits.hip with -DISG_EL_STAMP (s_memtime stamps per phase, no output depends on
them); the run launches it at the BASELINE configs[1] topology and prints the mean cycles per wave and phase."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "libisg_el_stamp.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                           "-DISG_EL_STAMP", os.path.join(CSRC, "isg_mp_logits.hip"), os.path.join(CSRC, "isg_graph.hip"),
                           "-o", OUT])
    print("built", OUT)
    sys.exit(0)

import torch

from isubgvqa_amd import ops, synthetic

lib = ctypes.CDLL(OUT)
c = ctypes
lib.isg_gatv2_edge_logits.argtypes = [c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int64, c.c_void_p,
                                      c.c_int32, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
                                      c.c_void_p, c.c_int64, c.c_int32, c.c_int32, c.c_int32, c.c_float, c.c_void_p]
lib.isg_el_set_stamp_buffer.argtypes = [c.c_void_p]
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
wgs = (E + 63) // 64
stamps = torch.zeros(wgs * 8, 8, dtype=torch.int64, device=dev)
assert lib.isg_el_set_stamp_buffer(stamps.data_ptr()) == 0
flush = torch.empty(1 << 27, device=dev)
st = torch.cuda.current_stream().cuda_stream
ts = []
for r in range(6):
    flush.fill_(float(r))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = lib.isg_gatv2_edge_logits(ea.data_ptr(), K, planes.data_ptr(), inv.data_ptr(), x_lr.data_ptr(), 2 * H * C, 0,
                                   x_lr.data_ptr() + 4 * H * C, 2 * H * C, 0, att.data_ptr(), plan.eid.data_ptr(),
                                   plan.src.data_ptr(), plan.dst.data_ptr(), None, None, lg.data_ptr(), E, H, C, K, 0.2, st)
    e.record()
    torch.cuda.synchronize()
    assert rc == 0
    ts.append(s.elapsed_time(e) * 1e3)
t = stamps.double().cpu()
active = t[t[:, 7] > 0]
names = ["staging (start -> panel barrier)", "issuing the gathers (per wave, all its tiles)", "k loops", "s_waitcnt vmcnt(0) after the k loop",
         "epilogue arithmetic", "flush + barrier + final reduction", "whole kernel (per wave)", "tiles per wave"]
print(f"isg_gatv2_edge_logits (stamped build) N={N} E={E} H={H} C={C} K={K}: {sorted(ts)[len(ts) // 2]:.1f} us per launch, "
      f"{wgs} workgroups x 8 waves, {active.size(0)} waves with tiles")
tot = active[:, 6].mean().item()
for i, n in enumerate(names):
    m = active[:, i].mean().item()
    print(f"  {n:48s} mean {m:10.0f}" + (f" cycles = {100 * m / tot:5.1f} % of the wave's time" if i < 6 else ""))
print(f"  cycles per tile: gathers issue {active[:, 1].sum().item() / active[:, 7].sum().item():.0f}, k loop "
      f"{active[:, 2].sum().item() / active[:, 7].sum().item():.0f}, gather wait {active[:, 3].sum().item() / active[:, 7].sum().item():.0f}, "
      f"epilogue {active[:, 4].sum().item() / active[:, 7].sum().item():.0f}")
