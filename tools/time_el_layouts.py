#!/usr/bin/env python3
"""isg_gatv2_edge_logits with x_l / x_r (a) as column halves of one [N, 2*H*C] tensor (4 KB row pitch: every gathered row in
its own page), (b) as separate dense [N, H*C] tensors (2 KB pitch), (c) head-major [2H][N][C] (512-byte rows, 8 per page).
Same values, same logits; HIP events, interleaved, 512 MiB written between launches.   python3 tools/time_el_layouts.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import _lib, ops, synthetic

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
K = wl.edge_attr.size(1)
HC = H * C
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
plan.require_csr()
g = torch.Generator(device=dev).manual_seed(0)
x_lr = torch.randn(N, 2 * HC, device=dev, generator=g)
xl_d, xr_d = x_lr[:, :HC].contiguous(), x_lr[:, HC:].contiguous()
hm = x_lr.view(N, 2 * H, C).permute(1, 0, 2).contiguous()          # [2H][N][C]: heads of x_l first, then of x_r
ea = wl.edge_attr.float().contiguous()
w = torch.randn(HC, K, device=dev, generator=g) / K ** 0.5
att = torch.randn(HC, device=dev, generator=g)
planes, inv = ops._weight_planes(w, True, "f16x3")
lib = _lib.load()
flush = torch.empty(1 << 27, device=dev)
layouts = {
    "a  column halves of [N, 2HC] (4 KB pitch)": (x_lr.data_ptr(), 2 * HC, 0, x_lr.data_ptr() + 4 * HC, 2 * HC, 0),
    "b  two dense [N, HC] tensors (2 KB pitch)": (xl_d.data_ptr(), HC, 0, xr_d.data_ptr(), HC, 0),
    "c  head-major [2H][N][C] (512 B rows)": (hm.data_ptr(), C, N * C, hm.data_ptr() + 4 * H * N * C, C, N * C),
}
outs, res = {}, {k: [] for k in layouts}
for r in range(rounds + 2):
    for name, (pl, ldl, hsl, pr, ldr, hsr) in layouts.items():
        lg = torch.empty(E, H, device=dev)
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = lib.isg_gatv2_edge_logits(ea.data_ptr(), K, planes.data_ptr(), inv.data_ptr(), pl, ldl, hsl, pr, ldr, hsr,
                                       att.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(), 0, 0,
                                       lg.data_ptr(), E, H, C, K, 0.2, torch.cuda.current_stream().cuda_stream)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        outs[name] = lg
        if r >= 2:
            res[name].append(s.elapsed_time(e) * 1e3)
ref = outs[next(iter(layouts))]
for name in layouts:
    v = sorted(res[name])
    print(f"{name:46s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us   logits equal to (a): {torch.equal(outs[name], ref)}")
