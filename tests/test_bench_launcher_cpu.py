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


def test_traffic_summaries_are_tied_to_kernel_kind_batch_shape_and_kernel_source(tmp_path):
    """roofline.traffic is REPLAYED from a committed PMC summary: it is taken only from a summary of the same kind of kernel
    (a pair's summary says nothing about the un-fused kernel), the same batch shape, and the same kernel SOURCE (sha256 written
    by tools/pmc_traffic.py): a summary measured before a kernel edit is refused, not silently kept."""
    import json
    import bench
    N, E = 82286, 205024
    good = {"kernel": "void isg::gatv2_layer_conv_kernel<false, 8>", "kind": "layer_conv", "N": N, "E": E,
            "hbm_bytes_per_launch": 330000000, "source_sha256": bench.kernel_source_hash("layer_conv")}
    json.dump(good, open(tmp_path / "x_mp_traffic.json", "w"))
    d = str(tmp_path)
    assert bench.load_traffic(N, E, "layer_conv", profiles_dir=d) == 330000000
    assert bench.load_traffic(N, E, "layer_conv", with_source=True, profiles_dir=d) == (330000000, "x_mp_traffic.json")
    assert bench.load_traffic(N + 1, E, "layer_conv", profiles_dir=d) is None          # another batch
    assert bench.load_traffic(N, E, "graph", profiles_dir=d) is None                   # another kernel
    stale = dict(good, source_sha256="0" * 64)
    json.dump(stale, open(tmp_path / "x_mp_traffic.json", "w"))
    got, why = bench.load_traffic(N, E, "layer_conv", with_source=True, profiles_dir=d)
    assert got is None and "another version" in why
    unstamped = {k: v for k, v in good.items() if k != "source_sha256"}
    json.dump(unstamped, open(tmp_path / "x_mp_traffic.json", "w"))
    assert bench.load_traffic(N, E, "layer_conv", profiles_dir=d) is None
    # the committed summary of this round's default kernel matches the committed source
    assert bench.load_traffic(N, E, "layer_conv") is not None


SAMPLE = "r06_bench_sample.json"        # a default-flag `python bench.py` line of this round, committed under profiles/ by name


def test_bench_line_schema_on_the_committed_sample():
    """The committed default-flag bench line of this round (profiles/r04_bench_sample.json: an explicit file, not "the newest
    by name") carries every object the contract and the review ask for: roofline, cpu_baseline (+cfg1), fallbacks,
    dense_err_vs_fp32, cfg5 (with the 8-rank imbalance of both partitions and the fp32-rows leg), full_model; a line without
    one of them must be rejected -- and so must a line whose `fallbacks` is None (what `--launch graph` produced in round 3)."""
    import copy
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.load(open(os.path.join(root, "profiles", SAMPLE)))
    bench.validate_line(line)
    assert line["metric"] == "GQA questions/sec" and line["config"]["graphs_per_gpu"] == 4096 and line["dtype"] == "f32"
    assert line["fallbacks"]["torch_linear"] == 0 and line["fallbacks"]["torch_layer_norm"] == 0
    assert 0 < line["dense_err_vs_fp32"]["max"] <= 2.0
    assert line["roofline"]["own_bound"] and "traffic_source" in line["roofline"]
    for drop in ("cfg5", "fallbacks", "roofline"):
        bad = copy.deepcopy(line)
        del bad[drop]
        try:
            bench.validate_line(bad)
        except KeyError:
            continue
        raise AssertionError(f"a line without {drop!r} passed validation")
    bad = copy.deepcopy(line)
    bad["fallbacks"] = None
    try:
        bench.validate_line(bad)
    except KeyError:
        pass
    else:
        raise AssertionError("a line with fallbacks = None passed validation")


def test_both_launch_modes_of_the_bench_count_their_fallbacks():
    """`--launch graph` takes its per-step counters from the eager steps it runs after the timed replays (round 3 left them
    None there and the finished run died in validate_line): every branch of main() that times steps assigns step_counters."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    branch = next(n for n in ast.walk(main) if isinstance(n, ast.If) and isinstance(n.test, ast.Compare)
                  and ast.unparse(n.test) == "args.launch == 'graph'" and any(isinstance(b, ast.For) for b in ast.walk(n)))
    assigns = lambda body: any(isinstance(t, ast.Name) and t.id == "step_counters" for st in body for n in ast.walk(st)
                               if isinstance(n, ast.Assign) for t in n.targets)
    assert assigns(branch.body), "the hipGraph branch never sets step_counters"
    assert assigns(branch.orelse), "the eager branch never sets step_counters"


def test_cfg5_workload_lines_name_configs4_and_carry_the_partition():
    """`bench.py --workload cfg5` (BASELINE configs[4] as the timed workload): the committed one-GPU line and the two-rank rehearsal
    (gloo ranks sharing the one MI355X) pass the schema of a run with the extra legs off, name configs[4], and the N > 1 line carries
    what the balanced partition gave every rank and the ragged collective's row counts."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = json.load(open(os.path.join(root, "profiles", "r06_a_cfg5_n1.json")))
    two = json.load(open(os.path.join(root, "profiles", "r06_a_cfg5_2rank_gloo.json")))
    for line in (one, two):
        bench.validate_line(line, full=False)
        assert "configs[4]" in line["config"]["workload"] and line["config"]["graphs_per_gpu"] == 2048
        assert line["summary"]["configs1_ms_per_step"] is None and line["summary"]["roofline_frac"] == line["roofline"]["frac"]
    assert one["n_gpus"] == 1 and one["config"]["global_batch"] == 2048
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 4096
    pr, rc = two["per_rank"], two["rccl"]
    assert sum(pr["graphs"]) == 4096 and pr["graphs"] == rc["rows_per_rank"] and rc["ragged"] and rc["padded_rows"] == max(pr["graphs"])
    assert 1.0 <= pr["imbalance_max_over_mean"] < 1.01 and len(pr["nodes_plus_edges"]) == 2
    args = bench.parse(["--workload", "cfg5"])
    assert (args.graphs, args.features) == (2048, "fp16")
    args = bench.parse([])
    assert (args.graphs, args.features, args.workload) == (4096, "fp32", "cfg2")
