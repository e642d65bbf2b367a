"""The RCCL calls of bench.py / distributed.py on a single-rank `nccl` group (one MI355X): two ranks cannot share a GPU under
NCCL, so the N > 1 path is rehearsed with gloo elsewhere (tests/test_distributed_cpu.py, bench.py's ISG_BENCH_BACKEND); this
checks that the RCCL library initialises on the box and that every collective call the multi-GPU run makes is accepted,
asynchronous handles included.  Runs in a subprocess so that the pytest process keeps no default process group."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    g = torch.Generator(device=dev).manual_seed(0)
    logits = torch.randn(4096, 1842, device=dev, generator=g)              # one rank's answer logits (30 MB)
    # distributed.GatherPipeline ITSELF, with bench.py's parsed default arguments, on RCCL (force_collective: a one-rank group
    # would otherwise skip the collective): two gathers in flight over three receive buffers
    import bench
    from isubgvqa_amd.distributed import GatherPipeline
    args = bench.parse([])
    pipe = GatherPipeline(logits.size(0), logits.size(1), dev, what=args.gather, depth=args.gather_depth, force_collective=True)
    assert pipe.active and len(pipe.buffers) == 3 and pipe.describe()["in_flight"] == 2
    outs, sent = [], []
    for i in range(4):
        outs.append(pipe.submit(i, logits))
        assert len(pipe.pending) <= 2
        sent.append(logits)
        logits = logits + 1.0                                               # next step's producer runs beside the collective
    pipe.drain()
    torch.cuda.synchronize()
    assert outs[3].data_ptr() == outs[0].data_ptr()
    assert torch.equal(outs[3], sent[3]) and torch.equal(outs[2], sent[2]) and torch.equal(outs[1], sent[1]), \
        "gathered logits differ from what was sent"
    # the ragged form (bench.py --workload cfg5): sizes exchanged once at construction, rows() trims
    a5 = bench.parse(["--workload", "cfg5"])
    rp = GatherPipeline(1000, 1842, dev, what=a5.gather, depth=a5.gather_depth, ragged=True, force_collective=True)
    assert rp.sizes == [1000] and rp.describe()["ragged"]
    part = sent[0][:1000].contiguous()
    got = rp.submit(0, part)
    rp.drain()
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(rp.rows(got)), part)
    # the opt-in light collective of bench.py (--gather answers): every rank's arg-max answers, i64, async like the logits
    ans = sent[3].argmax(dim=1)
    abuf = torch.empty_like(ans)
    w = dist.all_gather_into_tensor(abuf, ans, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    assert abuf.dtype == torch.int64 and torch.equal(abuf, ans)
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    devs = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=dev)
    out = [torch.empty_like(devs)]
    dist.all_gather(out, devs)
    assert int(out[0].item()) == 0
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL single-rank ok")
""") % ROOT


@pytest.mark.gpu
def test_rccl_collectives_of_the_multi_gpu_path_on_a_single_rank_group():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    err = [l for l in res.stderr.splitlines() if l.strip() and "amdgpu.ids" not in l]
    assert res.returncode == 0 and "RCCL single-rank ok" in res.stdout, "\n".join(err[-25:])
