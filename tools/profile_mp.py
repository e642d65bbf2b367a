#!/usr/bin/env python3
"""Run only the message-passing kernel on the BASELINE configs[1] batch (for rocprofv3 --pmc passes), plus a
known-size float4 copy used to calibrate FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/profile_mp.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/profile_mp.py
then  python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/rNN_mp_traffic.json
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "graph"      # graph | chunk | logits (the edge-logits pair) | tile (isg_gatv2_tile_conv)
ops.MP_KERNEL = "graph" if mode in ("logits", "tile", "layer") else mode
dev = torch.device("cuda:0")
cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs})
wl = synthetic.make_workload(cfg).to(dev)
N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
g = torch.Generator(device=dev).manual_seed(0)
x_l = torch.randn(N, H * C, device=dev, generator=g)
x_r = torch.randn(N, H * C, device=dev, generator=g)
e_proj = torch.randn(E, H * C, device=dev, generator=g)
att = torch.randn(1, H, C, device=dev, generator=g)
bias = torch.randn(H * C, device=dev, generator=g)
mask = (torch.rand(N, 1, device=dev, generator=g) > 0.7).float()
big = torch.randn(1 << 28, device=dev, generator=g)            # 1 GiB: larger than the 256 MiB Infinity Cache
dst = torch.empty_like(big)
torch.cuda.synchronize()
if mode in ("logits", "tile", "layer"):
    K = wl.edge_attr.size(1)
    ea = wl.edge_attr.float().contiguous()
    w = torch.randn(H * C, K, device=dev, generator=g) / K ** 0.5
    x_lr = torch.cat([x_l, x_r], 1).contiguous()                 # as the model has them: halves of one projection
    x_l, x_r = x_lr[:, :H * C], x_lr[:, H * C:]
    ops._weight_planes(w, True, "f16x3")                         # the split of W is not part of the pair
    torch.cuda.synchronize()
    fn = ops.gatv2_tile_conv if mode == "tile" else ops.gatv2_mp_edge_logits
    if mode == "layer":      # lin_l | lin_r inside: the operand is the gated layer input, not x_l / x_r
        from isubgvqa_amd.models.layers import GlorotLinear
        torch.manual_seed(0)
        lin_l, lin_r = GlorotLinear(C, H * C).to(dev), GlorotLinear(C, H * C).to(dev)
        xin = torch.randn(N, C, device=dev, generator=g)
        fn = lambda xl_, xr_, ea_, w_, att_, plan_, H_, **kw: ops.gatv2_layer_conv(xin, lin_l, lin_r, ea_, w_, att_, plan_, H_, **kw)
    if mode in ("tile", "layer"):
        plan.tiles(ops.TILE_CONV_NODES, ops.TILE_CONV_EDGES)      # the tile plan is built once per batch, not per layer
        torch.cuda.synchronize()
    for _ in range(reps):
        assert fn(x_l, x_r, ea, w, att, plan, H, bias=bias, want_rowmax=True) is not None
    for _ in range(reps):
        assert fn(x_l, x_r, ea, w, att, plan, H, bias=bias, node_mask=mask, want_rowmax=True) is not None
    if mode == "tile":       # own minimum: edge_attr + x_l + x_r in, out + alpha back, CSR (the logits never leave LDS)
        print(f"pair_own_bytes_unmasked={4 * E * K + 12 * N * H * C + 4 * E * H + 16 * E}")
    elif mode == "layer":    # edge planes + gated node rows in, out + alpha back, CSR
        print(f"pair_own_bytes_unmasked={4 * E * K + 4 * N * C + 4 * N * H * C + 4 * E * H + 16 * E}")
    else:
        print(f"pair_own_bytes_unmasked={ops.edge_logits_algorithmic_bytes(N, E, H, C, K, False) + ops.mp_logits_algorithmic_bytes(N, E, H, C, False)}")
else:
    for _ in range(reps):
        ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H, bias=bias)
    for _ in range(reps):
        ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H, bias=bias, node_mask=mask)
for _ in range(3):
    dst.copy_(big)                                              # calibration: reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
print(f"N={N} E={E} H={H} C={C} bytes_unmasked={ops.mp_algorithmic_bytes(N, E, H, C, False)} "
      f"bytes_masked={ops.mp_algorithmic_bytes(N, E, H, C, True)} copy_bytes={big.numel() * 4}")
