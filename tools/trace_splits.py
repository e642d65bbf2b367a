import sys, os, traceback, collections
sys.path.insert(0, os.getcwd())
import torch
from isubgvqa_amd import ops, synthetic
from isubgvqa_amd.models import build_model
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_model(synthetic.full_model_args(), None).to(dev).eval()
wl = synthetic.make_full_workload(512).to(dev)
sg = wl.scene_graphs()
ops.H3P_MIN_M = 1024
orig = ops.split_planes32
log = collections.Counter()
def spy(x, *a, **k):
    cached = getattr(x, "_isg_planes32", None) is not None
    st = traceback.extract_stack(limit=8)
    who = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in st[-5:-1][::-1])
    log[(tuple(x.shape), cached, who)] += 1
    return orig(x, *a, **k)
ops.split_planes32 = spy
with torch.no_grad():
    model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)
    log.clear()
    model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)
for (shape, cached, who), n in sorted(log.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1]):
    print(n, shape, "cached" if cached else "SPLIT", who)
