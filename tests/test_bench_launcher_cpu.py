"""`python bench.py --gpus N` must start its own ranks: the launcher is a child process running
torch.distributed.run, chosen before anything touches the GPU.  Reference launch shape: run_training_ddp.sh:23
(`torchrun --standalone --nproc_per_node=4`), main.py:72-94."""
import io
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_argv_shape():
    argv = bench.launcher_argv(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29511)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    i = argv.index("--master-addr")
    assert argv[i + 1] == "127.0.0.1"            # the container hostname may not resolve
    assert argv[argv.index("--master-port") + 1] == "29511"
    script = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]   # same flags, verbatim


class _FakeProc:
    def __init__(self, lines, rc):
        self.stdout = io.StringIO("".join(lines))
        self._rc = rc

    def wait(self):
        return self._rc


def test_self_launch_forwards_the_json_line_and_the_status(capsys, monkeypatch):
    seen = {}

    def fake_popen(cmd, stdout=None, text=None, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return _FakeProc(["W0000 some launcher chatter\n", '{"metric": "GQA questions/sec", "value": 1.0}\n'], 3)

    args = bench.parse(["--gpus", "4", "--steps", "2"])
    rc = bench.self_launch(args, ["--gpus", "4", "--steps", "2"], run=fake_popen)
    out = capsys.readouterr()
    assert rc == 3
    assert out.out == '{"metric": "GQA questions/sec", "value": 1.0}\n'     # stdout carries the ONE JSON line
    assert "launcher chatter" in out.err
    assert "--nproc-per-node=4" in seen["cmd"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_is_an_error(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert "does not match" in str(e.value)


def test_main_self_launches_when_no_launcher_is_present(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    called = {}

    def fake_self_launch(args, argv):
        called["gpus"], called["argv"] = args.gpus, argv
        return 0

    monkeypatch.setattr(bench, "self_launch", fake_self_launch)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "1"])
    assert e.value.code == 0 and called == {"gpus": 2, "argv": ["--gpus", "2", "--steps", "1"]}


@pytest.mark.timeout(300)
def test_real_launch_reaches_the_ranks_and_returns_their_status():
    """No GPU here: each rank must come up under torch.distributed.run with WORLD_SIZE=2 and stop at the product path's
    'needs an MI355X' assertion; the parent must report the failure, not hang or succeed."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the real run is bench.py's job")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr
    assert p.stdout.strip() == ""


def test_cpu_legs_use_the_cpus_the_process_may_run_on_and_are_time_bounded():
    """The configs[0] leg once took os.cpu_count() threads (the HOST's count) and a minimum of seven passes: on a GPU box
    with a 16-CPU share it ran past seven silent minutes and the run was killed.  It takes the usable CPUs and one warm +
    at least one timed pass now."""
    import os
    import time
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    r = bench.cpu_baseline_cfg1(seconds=0.5)
    took = time.perf_counter() - t0
    assert r["cores"] == min(n, 32) and r["value"] > 0 and r["kind"] == "port"
    assert took < 60, f"configs[0] leg took {took:.1f} s for a 0.5 s budget"


def test_traffic_summaries_are_replayed_only_for_the_kernels_they_describe():
    """bench.py replays committed PMC summaries into roofline.traffic.  The edge-logits pair's summary (two kernels summed,
    874 MB) was once written without its kind and came back as the un-fused per-graph kernel's traffic (943 MB): every
    summary is matched by what was summed, and the three committed kinds stay apart."""
    import bench
    N, E = 82286, 205024
    pair, graph, chunk = (bench.load_traffic(N, E, k) for k in ("logits_pair", "graph", "chunk"))
    assert pair is not None and graph is not None and chunk is not None and len({pair, graph, chunk}) == 3
    assert 0.9e9 < graph < 1.0e9 and pair < graph < chunk
    assert bench.load_traffic(N + 1, E, "graph") is None


def test_bench_line_schema_on_the_committed_sample():
    """The newest committed default-flag bench line of this round (profiles/r03_*bench.json) carries every object the contract
    and the review ask for: roofline, cpu_baseline (+cfg1), fallbacks, dense_err_vs_fp32, cfg5 (with the 8-rank imbalance of
    both partitions), full_model; a line without one of them must be rejected."""
    import copy
    import glob
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r03_*_bench.json")))
    assert paths, "no committed r03 bench line under profiles/"
    line = json.load(open(paths[-1]))
    bench.validate_line(line)
    assert line["metric"] == "GQA questions/sec" and line["config"]["graphs_per_gpu"] == 4096 and line["dtype"] == "f32"
    assert line["fallbacks"]["torch_linear"] == 0 and line["fallbacks"]["torch_layer_norm"] == 0
    assert 0 < line["dense_err_vs_fp32"]["max"] <= 2.0
    for drop in ("cfg5", "fallbacks", "roofline"):
        bad = copy.deepcopy(line)
        del bad[drop]
        try:
            bench.validate_line(bad)
        except KeyError:
            continue
        raise AssertionError(f"a line without {drop!r} passed validation")
