"""Time one training step (forward + backward, no optimizer) of the cfg2 workload on the HIP path.
Usage: python tools/time_train.py [--graphs 4096] [--steps 10] [--sampler gumbel|imle|aimle]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isubgvqa_amd  # noqa: E402,F401
from isubgvqa_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sampler", default="gumbel")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = synthetic.WorkloadConfig(num_graphs=a.graphs, sampler=a.sampler)
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).train()
    target = torch.randint(0, 1842, (cfg.num_graphs,), device=dev)

    def step(i):
        model.zero_grad(set_to_none=True)
        logits, _, _ = model(wl, seed=1000 + i)
        loss = torch.nn.functional.cross_entropy(logits, target)
        loss.backward()
        return loss

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    with torch.no_grad():
        model.eval()
        for i in range(a.warmup):
            model(wl, seed=i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            model(wl, seed=i)
        torch.cuda.synchronize()
        df = (time.perf_counter() - t0) / a.steps
    print(f"train step (fwd+bwd) {dt * 1e3:.2f} ms  = {cfg.num_graphs / dt:,.0f} questions/s; "
          f"inference forward {df * 1e3:.2f} ms; loss {float(loss):.4f}")


if __name__ == "__main__":
    main()
