#!/usr/bin/env python3
"""Run the three tile kernels of a BASELINE configs[1] layer a few times each (for rocprofv3 --pmc passes):
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
            SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/x -- python3 tools/profile_tile_kernels.py
then python3 tools/pmc_sum.py <kernel> gpurun_out/x"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isubgvqa_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
net = synthetic.build_answer_model(cfg).to(dev).eval()
m = net.gat_seq
conv = m.convs[0]
N, H, C = wl.x.size(0), cfg.heads, cfg.channels
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
x = wl.x.contiguous()
ins, ins_next = wl.instr[0].contiguous(), wl.instr[1].contiguous()
bn = m.bns[0]
with torch.no_grad():
    for rep in range(5):
        out, _ = ops.gatv2_layer_conv(x, conv.lin_l, conv.lin_r, wl.edge_attr, conv.lin_edge.weight, conv.att, plan, H, bias=conv.bias,
                                      want_rowmax=True)
        h, xg, _ = ops.mgat_dense_tail(out, m.x_proj[0], ins, x, plan, bn.weight, bn.bias, bn.mean_scale, bn.eps, ins_next=ins_next)
        ops.readout_tile(h, net.graph_global_attention_pooling.node_nn, wl.glf, plan)
    if "--tile-conv" in sys.argv:
        # round 6: isg_gatv2_tile_conv (x_l / x_r projected before it) -- the SAME edge chunks, logit epilogue and aggregation as the
        # layer kernel, as TWO independent four-wave workgroups per CU (74 KB of LDS each) instead of one eight-wave workgroup: the
        # in-tree instance of "two independent pipelines per CU" 
        x_l, x_r = ops.linear_fused(x, (conv.lin_l, conv.lin_r))
        for rep in range(5):
            ops.gatv2_tile_conv(x_l, x_r, wl.edge_attr, conv.lin_edge.weight, conv.att, plan, H, bias=conv.bias, want_rowmax=True)
torch.cuda.synchronize()
print("ok")
