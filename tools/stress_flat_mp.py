#!/usr/bin/env python3
"""The flat message-passing kernel (C = 300, the full model's) launched twice on the same inputs, in a loop: same bits every
time?  (tools/scan_pk_waw.py finds in it the operand-selection pattern of the tile conv's intermittent wrong sums.)
  python3 tools/stress_flat_mp.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from isubgvqa_amd import ops
from test_gpu_ops import _rand_graphs

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(31)
H, C = 4, 300
bad = 0
for it in range(iters):
    sizes = torch.randint(8, 34, (700,), generator=gen).tolist()
    batch, ei = _rand_graphs(gen, sizes, extra_per_node=1.5, hub=(7, 60))
    N, E, B = batch.numel(), ei.size(1), len(sizes)
    x_l, x_r = torch.randn(N, H * C, generator=gen).to(dev), torch.randn(N, H * C, generator=gen).to(dev)
    e_proj = torch.randn(E, H * C, generator=gen).to(dev)
    att, bias = torch.randn(1, H, C, generator=gen).to(dev), torch.randn(H * C, generator=gen).to(dev)
    masked = it % 2 == 1
    nm = (torch.rand(N, generator=gen) < 0.7).float().to(dev) if masked else None
    plan = ops.GraphPlan.build(batch.to(dev), ei.to(dev), num_graphs=B)
    with torch.no_grad():
        r = [ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H, bias=bias, node_mask=nm) for _ in range(3)]
    torch.cuda.synchronize()
    for j in (1, 2):
        if not (torch.equal(r[j][0], r[0][0]) and torch.equal(r[j][1], r[0][1])):
            bad += 1
            d = r[j][0] != r[0][0]
            rows = d.any(1).nonzero().flatten()
            print(f"iteration {it} launch {j}: {int(d.sum())} values in {rows.numel()} rows, columns {d[rows[0]].nonzero().flatten().tolist()[:20]}, "
                  f"max |d| {(r[j][0] - r[0][0]).abs().max().item():.3g}; alpha equal {torch.equal(r[j][1], r[0][1])}", flush=True)
print(f"{bad} of {2 * iters} repeated launches differ")
