#!/usr/bin/env python3
"""Mixed dispatch on / off on a configs[1] batch with a few graphs beyond a tile (the case real GQA batches are: the reference
caps nothing, datasets/scene_graph.py:199-389).   python3 tools/time_mixed.py [graphs] [big graphs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isubgvqa_amd import ops, synthetic

only = None          # --only=split / --only=fill / --only=off: one mode (for a rocprofv3 kernel trace of it)
for a in list(sys.argv[1:]):
    if a.startswith("--only="):
        only = a.split("=", 1)[1]
        sys.argv.remove(a)
sync_debug = "--syncdebug" in sys.argv        # warn at every device-to-host synchronisation of a step
if sync_debug:
    sys.argv.remove("--syncdebug")
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
ops.MIXED_MAX_FRACTION, ops.MIXED_MIN_NODES = 0.9, 0      # the gate open: this tool measures the mode itself
gen = torch.Generator().manual_seed(3)
for nbig in ([int(sys.argv[2])] if len(sys.argv) > 2 else [0, 1, 8, 64]):
    base = synthetic.graph_sizes(synthetic.WorkloadConfig(num_graphs=graphs), gen).tolist()
    for i in range(nbig):
        base[(i * 977 + 13) % graphs] = 100 + (i * 37) % 90
    cfg = synthetic.WorkloadConfig(num_graphs=graphs, sizes=tuple(base))
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    line = f"{graphs} graphs, {nbig} beyond a tile (N={wl.x.size(0)}, max nodes {wl.max_nodes}, max edges {wl.max_edges}):"
    # split: the graphs beyond a tile as a batch of their own through the whole model (ops.run_split); fill: every tile kernel's
    # wrapper fills their rows (round 4's first form); off: the per-graph kernels for the whole batch
    # split2: split with the sub-batch on a stream of its own (ops.SPLIT_STREAM, the default)
    for mode in (("split2", "split", "fill", "off", "split2", "split", "fill", "off") if only is None else (only,)):
        ops.MIXED_DISPATCH = mode != "off"
        ops.SPLIT_FORWARD = mode.startswith("split")
        ops.SPLIT_STREAM = mode == "split2"
        with torch.no_grad():
            for i in range(3):
                model(wl, seed=50 + i)
            torch.cuda.synchronize()
            if sync_debug:
                torch.cuda.set_sync_debug_mode("warn")
                model(wl, seed=59)
                torch.cuda.set_sync_debug_mode("default")
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                model(wl, seed=60 + i)
            t_issue = (time.perf_counter() - t0) / 20          # the host's share: launches issued, nothing waited for
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
        line += f"  {mode} {dt * 1e3:.3f} ms (host issue {t_issue * 1e3:.3f})"
    print(line, flush=True)
