import sys, os
sys.path.insert(0, os.getcwd())
import torch
from isubgvqa_amd import ops, synthetic
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
wl = synthetic.make_workload(cfg).to(dev)
plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
tp, nt, cap, info = plan.tiles(64, 256)
T = int(nt.item())
info = info[:T].cpu()
nodes, slots = info[:, 1].float(), info[:, 3].float()
chunks = torch.ceil(slots / 64)
half = ((slots - 64 * (chunks - 1)) <= 32).float()          # last chunk's upper half idle
# cycles per (tile, head) from the stamps: node GEMM ~4.9k fixed, per chunk ~3.6k (a half chunk ~2.6k), aggregation ~ 50 per slot
cost = 5000 + 3600 * chunks - 1000 * half + 22 * slots + 20 * nodes
print("tiles", T, "mean cost", cost.mean().item(), "std/mean", (cost.std() / cost.mean()).item(), "chunks hist", torch.bincount(chunks.long()).tolist())
G = 64
def spread(order):
    per = torch.zeros(G)
    for w in range(G):
        per[w] = cost[order[w::G]].sum()
    return (per.max() / per.mean()).item()
print("static order     : max/mean per workgroup", spread(torch.arange(T)))
print("sorted by cost   : ", spread(torch.argsort(cost, descending=True)))
print("bucket by chunks : ", spread(torch.argsort(chunks, descending=True, stable=True)))

import heapq
def lpt(order):          # dynamic fetch: the next tile of the list goes to the workgroup that is free first
    h = [0.0] * G
    heapq.heapify(h)
    for t in order.tolist():
        heapq.heappush(h, heapq.heappop(h) + cost[t].item())
    return max(h) / (cost.sum().item() / G)
print("dynamic fetch, batch order    :", lpt(torch.arange(T)))
print("dynamic fetch, heavy first    :", lpt(torch.argsort(chunks, descending=True, stable=True)))
