import sys, time, torch
sys.path.insert(0, "/root/repo")
from isubgvqa_amd import ops, synthetic
from isubgvqa_amd.models import build_model
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(synthetic.full_model_args(), None).to(dev).eval()
for T in (12, 16, 17, 20):
    wl = synthetic.make_full_workload(8, tokens=T).to(dev)
    sg = wl.scene_graphs()
    f = lambda **kw: model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg, **kw)
    with torch.no_grad():
        for mode in ({}, {"capture": True}):
            for _ in range(4): f(**mode)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): f(**mode)
            torch.cuda.synchronize()
            print(T, mode, round((time.perf_counter() - t0) / 30 * 1e3, 3), "ms", flush=True)
