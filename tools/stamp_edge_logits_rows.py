#!/usr/bin/env python3
"""Where does a wave of isg_gatv2_edge_logits' rows kernel (C = 300, K = 300) spend its cycles?  `--build` (in the build
container) makes tools/_build/libisg_er_stamp.so from isg_mp_logits.hip with -DISG_DIAG; the run launches it once at the BASELINE
configs[1] topology (4096 graphs by default) and prints the stamps of the computing waves and of the requesting wave.
    python3 tools/stamp_edge_logits_rows.py --build
    python3 tools/stamp_edge_logits_rows.py [graphs]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "libisg_er_stamp.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DISG_DIAG", "-fno-slp-vectorize",
                           *[a for a in sys.argv[1:] if a.startswith("-D")],
                           os.path.join(CSRC, "isg_mp_logits.hip"), os.path.join(CSRC, "isg_graph.hip"), "-o", OUT])
    print("built", OUT)
    sys.exit(0)

import torch

from isubgvqa_amd import _lib, ops, synthetic

stamp = ctypes.CDLL(OUT)
stamp.isg_gatv2_edge_logits.restype, stamp.isg_gatv2_edge_logits.argtypes = _lib.SIGNATURES["isg_gatv2_edge_logits"]
stamp.isg_er_set_stamp_buffer.argtypes = [ctypes.c_void_p]
args = [a for a in sys.argv[1:] if not a.startswith("-")]
graphs = int(args[0]) if args else 4096
dev = torch.device("cuda:0")
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs, "channels": 300})
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C, K = wl.x.size(0), wl.edge_index.size(1), cfg.heads, 300, 300
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_lr = torch.randn(N, 2 * H * C, device=dev, generator=g)
ea = wl.edge_attr.float()[:, :K].contiguous()
w = torch.randn(H * C, K, device=dev, generator=g) / K ** 0.5
att = torch.randn(H * C, device=dev, generator=g)
wf, winv = ops._edge_logits_weight(w, H)
logits = torch.empty(E, H, device=dev)
nwg = (E + 223) // 224
buf = torch.zeros(nwg * 8, 16, dtype=torch.int64, device=dev)
assert stamp.isg_er_set_stamp_buffer(buf.data_ptr()) == 0
for rep in range(2):
    buf.zero_()
    rc = stamp.isg_gatv2_edge_logits(ea.data_ptr(), K, wf.data_ptr(), winv.data_ptr(), x_lr.data_ptr(), 2 * H * C, 0,
                                     x_lr[:, H * C:].data_ptr(), 2 * H * C, 0, att.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(),
                                     plan.dst.data_ptr(), None, None, logits.data_ptr(), E, H, C, K, 0.2,
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
s = buf.double().cpu().view(nwg, 8, 16)
comp, req = s[:, :7].reshape(-1, 16), s[:, 7]
tiles = H * 320 // 32
for what, t, names in (("computing waves", comp, ["rows -> planes (once)", "tile barrier", "gathers: addresses + requests", "k loop (57 products)",
                                                  "wait for the gathers", "epilogue", "", "", "", "", "", "", "(total)"]),
                       ("requesting wave", req, ["first two tiles' requests (once)", "counted wait: the tile has landed", "tile barrier",
                                                 "requests of tile nt + 2", "", "", "", "", "", "", "", "", "(total)"])):
    tot = t[:, 12].mean().item()
    print(f"{what}: {t.size(0)} waves, {nwg} workgroups, {tiles} tiles; a wave lives {tot:.0f} cycles = {tot / tiles:.0f} per tile")
    for i, n in enumerate(names):
        if n:
            m = t[:, i].mean().item()
            print(f"  {n:45s} {m:10.0f}  ({100 * m / tot:5.1f} %)   per tile {m / tiles:8.0f}   max {t[:, i].max().item():10.0f}")
