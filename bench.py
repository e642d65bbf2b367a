#!/usr/bin/env python
"""Benchmark of the ISubGVQA inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N>1 runs one rank per GPU under torch.distributed.run (RCCL).  Started WITHOUT a launcher (WORLD_SIZE unset) the
script starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` itself as a CHILD
process -- before anything touches the GPU, never by exec -- forwards rank 0's JSON line and exits with the child's
status (the reference's launch shape: run_training_ddp.sh:23 `torchrun --standalone --nproc_per_node=4`, main.py:72-94).
Started by a launcher (WORLD_SIZE set) it is a rank; WORLD_SIZE != --gpus is an error.

A step = one pass of the hot path over one resident batch: BASELINE.json configs[1]
(4096 synthetic GQA-shaped scene graphs per GPU, ~20 nodes / ~50 edges, 3 masked-GATv2 layers at C=128, H=4,
Gumbel top-k k=5, then attention pooling and the 1842-way classifier), i.e. ISubGVQA.forward from
`gat_seq` down (ISubGVQA/models/isubgvqa.py:267-292) including the per-batch graph plan (CSR build).  Inputs
are in HBM before the timed region.  With N ranks every rank owns its own 4096-graph shard (weak scaling) and
each step ends with the RCCL all-gather of answer logits [B_local, 1842] fp32 -- the only collective of the path (BASELINE
north_star / configs[3]; --gather answers moves the arg-max answers instead and says so in rccl.collective); up to
--gather-depth (2) of them are in flight on the communicator's stream beside the next steps' kernels (the last ones are
waited for inside the timed region).

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      the reference's message + aggregate: algorithmic bytes (e_proj included) / mean duration measured
                with HIP events on the launch stream over the timed region, against the HBM peak.  By default that function
                runs as TWO launches with lin_edge inside the first (isg_gatv2_edge_logits + isg_gatv2_mp_fwd_logits): the
                bracket covers both, `parts` gives each against its own bytes, `unfused_kernel` the round-1 kernel timed in the
                same run outside the timed region; --no-fuse-logits times the round-1 boundary as the step.  `traffic` is
                replayed from the committed PMC summary of exactly those kernels (profiles/*_mp_traffic.json)
  cpu_baseline  the CPU oracle (oracle/model.py, a port of the reference's PyG CPU path) timed on this host's usable
                cores on a bounded sample of the same workload; .cfg1 = BASELINE configs[0] exactly
  full_model    the configs[2] stand-in (full model at C = 300) timed in the same run; .kernels = its two dominant kernels against
                their rooflines from HIP events around every launch of three extra steps (isg_linear_h3p: TFLOP/s of fp16
                products vs the 2.5 PF dense MFMA peak; the message-passing pair -- edge logits + flat kernel from logits --: SURVEY 8(d)'s
                bytes at H C = 1200 vs HBM)
  sustained     >= 2.5 s of back-to-back configs[1] steps behind the timed burst: ms/step and its ratio to the burst's
  cfg5          BASELINE configs[4] on one GPU (skewed graphs, AIMLE, fp16 rows): ms/step, MP kernel GB/s on s = 2 bytes, the
                imbalance of contiguous graph ranges over 8 ranks
  eval_batch    the reference's evaluation batch (1 024 graphs) on the product path: eager vs its hipGraph option (capture=True)
  small_batch   the full model at 1 / 8 / 32 questions (run_token_coo.py's batch is one): eager and capture=True
  summary       the legs' headline numbers, flat, at the front of the line
  mixed         the configs[1] batch with a few graphs beyond a graph tile (what real GQA batches are: the reference caps nothing):
                ms/step with the tile kernels + the big graphs as a sub-batch (ops.run_split) and with the per-graph kernels for all
  fallbacks     launches per step that left this library's dense kernels (0 = none), extra row-maximum passes
  dense_err_vs_fp32   error of the exact-split dense kernels relative to a plain fp32 GEMM's (vs fp64)
  per_rank      (N > 1) every rank's own ms/step and its wait at the closing barrier
  rccl          backend, world size, every rank's device index
--launch graph replays the step as one captured hipGraph (single GPU; pays below ~500 graphs per step).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBPS = 6290.0
MFMA_F16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA, ~2.5 PF (never the 2:1-sparsity headline)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--graphs", type=int, default=None,
                    help="graphs per GPU (default: 4096 = BASELINE configs[1]; 2048 with --workload cfg5)")
    ap.add_argument("--workload", choices=["cfg2", "cfg5"], default="cfg2",
                    help="cfg2 (default): BASELINE configs[1] / configs[3], every rank its own 4096-graph batch.  cfg5: BASELINE "
                         "configs[4] -- ONE generated batch of N x --graphs skewed graphs (8-200 nodes, power-law in-degree, AIMLE k=5, "
                         "fp16 feature rows) sharded into contiguous graph ranges balanced by sum(nodes + edges) "
                         "(distributed.shard_workload(balance=True)); ranks then hold different graph counts and the logits "
                         "all-gather is the ragged GatherPipeline (row counts exchanged once, rows padded to the largest shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the configs[4] leg (skewed graphs, AIMLE, fp16 rows)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-dispatch leg (configs[1] + a few graphs beyond a tile)")
    ap.add_argument("--no-full-model", action="store_true", help="skip the configs[2] stand-in (full model at C = 300)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained leg (>= --sustained-seconds of steps)")
    ap.add_argument("--sustained-seconds", type=float, default=2.5)
    ap.add_argument("--full-model-graphs", type=int, default=4096)
    ap.add_argument("--cpu-sample-graphs", type=int, default=512)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch threads for the CPU baseline (capped at the host's)")
    ap.add_argument("--no-hints", action="store_true", help="let the plan read Nmax back from the device (one sync)")
    ap.add_argument("--mp-kernel", choices=["graph", "chunk"], default="graph")
    ap.add_argument("--gemm", choices=["bf16x6", "torch"], default="bf16x6")
    ap.add_argument("--launch", choices=["eager", "graph"], default="eager",
                    help="graph: the step (plan build included) captured once as a hipGraph and replayed; Gumbel noise from "
                         "torch's generator inside the graph (fresh on every replay); single GPU")
    ap.add_argument("--gather", choices=["logits", "answers"], default="logits",
                    help="N > 1: what every rank receives per step -- the fp32 logits [B_local,1842] of every peer (BASELINE "
                         "north_star / configs[3]: 30 MB per rank per step at 4096 graphs, 211 MB received at N = 8; the default), "
                         "or only their arg-max answers [B_local] i64 (32 KB per rank: what the reference's evaluation loop "
                         "needs, utils/misc.py:40-48; opt-in, reported as such in rccl.collective)")
    ap.add_argument("--gather-depth", type=int, default=2,
                    help="N > 1: all-gathers in flight (depth + 1 receive buffers): depth 2 hides a collective of up to two "
                         "steps' length behind the following steps' kernels (DESIGN 7, option 2)")
    ap.add_argument("--no-fuse-logits", action="store_true",
                    help="A/B: lin_edge as its own GEMM + the message-passing kernel streaming e_proj (the round-1 boundary)")
    ap.add_argument("--features", choices=["fp32", "fp16"], default=None,
                    help="storage of the projected rows (default: fp32; fp16 with --workload cfg5 = BASELINE configs[4]'s storage)")
    args = ap.parse_args(argv)
    if args.graphs is None:
        args.graphs = 4096 if args.workload == "cfg2" else 2048
    if args.features is None:
        args.features = "fp32" if args.workload == "cfg2" else "fp16"
    return args


def launcher_argv(gpus: int, script_args, port: int):
    """Command line of the child that runs this script as `gpus` ranks on one node (rendezvous on 127.0.0.1: the
    container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *script_args]


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, script_args, run=subprocess.Popen) -> int:
    """--gpus N>1 without a launcher: start torch.distributed.run as a child process (this process has not imported
    torch, let alone initialised the GPU), pass rank 0's JSON line through on stdout, everything else on stderr, and
    return the child's exit status."""
    cmd = launcher_argv(args.gpus, script_args, _free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        if line.lstrip().startswith('{"metric"'):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def cpu_baseline(cfg, sample_graphs: int, seconds: float, threads: int):
    """Oracle (kind 'port') on the host cores: same workload distribution, `sample_graphs` graphs per pass."""
    import torch
    from isubgvqa_amd import synthetic
    from oracle import model as OM
    from oracle import samplers as OS
    scfg = synthetic.WorkloadConfig(**{**cfg.__dict__, "num_graphs": sample_graphs})
    wl = synthetic.make_workload(scfg)
    model = synthetic.build_answer_model(scfg).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ocfg = OM.PathConfig(heads=scfg.heads, masking_thresholds=list(scfg.masks), use_topk=True,
                         sampler_type=scfg.sampler, sample_k=scfg.sample_k)
    gen = torch.Generator().manual_seed(1)
    noises = {i: OS.uniform_to_gumbel(torch.rand(sample_graphs, wl.max_nodes, generator=gen))
              for i, t in enumerate(scfg.masks) if t != 1.0}
    threads = max(1, min(threads, usable_cpus()))
    torch.set_num_threads(threads)       # many-core hosts: small graph ops slow down past a few dozen threads
    with torch.no_grad():
        t0 = time.perf_counter()
        OM.mgat_pool_classify(sd, wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.instr, wl.glf, ocfg, noises)
        progress(f"cpu_baseline: warm pass over {sample_graphs} graphs {time.perf_counter() - t0:.2f} s on {threads} threads")
        t0 = time.perf_counter()
        passes = 0
        while passes < 1 or (time.perf_counter() - t0 < seconds and passes < 200):
            OM.mgat_pool_classify(sd, wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.instr, wl.glf, ocfg, noises)
            passes += 1
        dt = time.perf_counter() - t0
    return {"value": round(sample_graphs * passes / dt, 1), "unit": "questions/s", "cores": threads, "kind": "port",
            "sample": f"{passes} passes x {sample_graphs} graphs of the configs[1] distribution "
                      f"(N={wl.x.size(0)}, E={wl.edge_index.size(1)}) in {dt:.1f} s, torch CPU fp32, "
                      f"{threads} threads ({usable_cpus()} CPUs usable by the process, host has {os.cpu_count()})"}


def cpu_model_string() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus() -> int:
    """CPUs this process may actually run on: the affinity mask, cut by the cgroup's CPU quota when there is one.
    os.cpu_count() is the HOST's count; on a GPU box with a 16-CPU share, that many OpenMP threads over 32-graph batches
    is an oversubscribed crawl (round 1: 390 q/s on 128 threads, 1638 q/s on 16), and it made this leg run past 7 minutes."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def progress(msg: str):
    """One line per leg on stderr (rank 0's JSON line stays the only thing on stdout): a run that is killed for
    silence or at its limit says where it was."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline_single_question(seconds: float = 4.0, max_threads: int = 16):
    """The CPU path's latency for ONE question through the FULL model (run_token_coo.py:49-79 evaluates one question per forward): the
    oracle's isubgvqa_forward (kind 'port') on the same synthetic question / scene graph the `small_batch` leg times on the GPU,
    median of the passes that fit `seconds`, at 1 thread and at `max_threads` (a 21-node graph does not scale with threads)."""
    import torch
    from isubgvqa_amd import synthetic
    from isubgvqa_amd.models import build_model
    from oracle import model as OM
    torch.manual_seed(0)
    model = build_model(synthetic.full_model_args(), None).eval()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    wl = synthetic.make_full_workload(1)
    ocfg = OM.PathConfig(heads=4, masking_thresholds=[1.0, 1.0, 1.0, 0.15], sampler_type="imle", sample_k=5)
    out = {"unit": "ms per question", "kind": "port", "what": "oracle.model.isubgvqa_forward, 1 question of 12 tokens, full model at C = 300"}
    with torch.no_grad():
        for threads in (1, min(usable_cpus(), max_threads)):
            torch.set_num_threads(threads)
            f = lambda: OM.isubgvqa_forward(sd, wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, wl.x_bbox,
                                            wl.added_sym_edge, ocfg, None)
            f()
            times, t_end = [], time.perf_counter() + seconds / 2
            while not times or (time.perf_counter() < t_end and len(times) < 200):
                t0 = time.perf_counter()
                f()
                times.append(time.perf_counter() - t0)
            out[f"threads_{threads}_ms"] = round(sorted(times)[len(times) // 2] * 1e3, 3)
    return out


def cpu_baseline_cfg1(seconds: float = 8.0, max_threads: int = 32):
    """SURVEY §8(d) / BASELINE configs[0] exactly: 256 graphs as 8 batches of 32 (<= 16 nodes, <= 32 edges), C = 300, 4 MGAT
    layers, masks [1, 1, 1, 0.15], Gumbel k = 5 with explicit noise, the oracle under no_grad on every CPU this process
    may use (`usable_cpus`, at most `max_threads`).  Bounded: one warm pass, then timed passes until `seconds` are spent
    (at least one, so a slow host costs one pass, not five)."""
    import torch
    from isubgvqa_amd import synthetic
    from oracle import model as OM
    from oracle import samplers as OS
    threads = min(usable_cpus(), max_threads)
    torch.set_num_threads(threads)
    batches = []
    for b in range(8):
        scfg = synthetic.WorkloadConfig(**{**synthetic.CFG1.__dict__, "seed": synthetic.CFG1.seed + b})
        wl = synthetic.make_workload(scfg)
        gen = torch.Generator().manual_seed(100 + b)
        noises = {i: OS.uniform_to_gumbel(torch.rand(scfg.num_graphs, wl.max_nodes, generator=gen))
                  for i, t in enumerate(scfg.masks) if t != 1.0}
        batches.append((wl, noises))
    model = synthetic.build_answer_model(synthetic.CFG1).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ocfg = OM.PathConfig(heads=4, masking_thresholds=list(synthetic.CFG1.masks), use_topk=True, sampler_type="gumbel",
                         sample_k=synthetic.CFG1.sample_k)

    def one_pass():
        for wl, noises in batches:
            OM.mgat_pool_classify(sd, wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.instr, wl.glf, ocfg, noises)

    times = []
    with torch.no_grad():
        t0 = time.perf_counter()
        one_pass()
        progress(f"cpu_baseline.cfg1: warm pass {time.perf_counter() - t0:.2f} s on {threads} threads")
        t_end = time.perf_counter() + seconds
        while not times or (time.perf_counter() < t_end and len(times) < 100):
            t0 = time.perf_counter()
            one_pass()
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(256 / med, 1), "unit": "questions/s", "cores": threads, "kind": "port",
            "cpu": cpu_model_string(),
            "sample": f"BASELINE configs[0]: 256 graphs = 8 batches x 32 (<= 16 nodes, <= 32 edges), C=300, 4 layers, "
                      f"masks [1,1,1,0.15], Gumbel k=5; median of {len(times)} passes ({med * 1e3:.1f} ms/pass), "
                      f"torch CPU fp32, {threads} threads = the CPUs usable by the process "
                      f"(host has {os.cpu_count()} logical cores)"}


def full_model_rate(dev, graphs: int, steps: int = 10):
    """BASELINE configs[2] stand-in, measured in the same run: the FULL model (question encoder/decoder, scene-graph
    encoder, 4 MGAT layers at C = 300, I-MLE k = 5, pooling, classifier) on GQA-shaped synthetic token batches."""
    import torch
    from isubgvqa_amd import ops, synthetic
    from isubgvqa_amd.models import build_model
    torch.manual_seed(0)
    model = build_model(synthetic.full_model_args(), None).to(dev).eval()
    wl = synthetic.make_full_workload(graphs).to(dev)
    sg = wl.scene_graphs()
    with torch.no_grad():
        for _ in range(3):
            out = model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True,
                        scene_graphs=sg)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True,
                        scene_graphs=sg)[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out).all()
    # the step's two dominant kernels against THEIR rooflines, from HIP events around every launch of three more steps (outside
    # the timed loop: an event pair per launch costs host time): isg_linear_h3p against the dense fp16 MFMA peak (three fp16
    # products per fp32 product: 6 M N K flops per launch), the message-passing pair against HBM on SURVEY 8(d)'s bytes_mp
    ops.H3P_TIMER, ops.MP_TIMER = ops.KernelTimer(), ops.KernelTimer()
    with torch.no_grad():
        for _ in range(3):
            model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True, scene_graphs=sg)
    torch.cuda.synchronize()
    ht, mt = ops.H3P_TIMER, ops.MP_TIMER
    ops.H3P_TIMER, ops.MP_TIMER = None, None
    hd, md = ht.durations_ms(), mt.durations_ms()
    flops = [6.0 * m["M"] * m["N"] * m["K"] for m in ht.meta]
    k300 = [i for i, m in enumerate(ht.meta) if m["K"] <= 320]
    mpb = [ops.mp_algorithmic_bytes(m["N"], m["E"], m["H"], m["C"], m["masked"], m.get("feat_bytes", 4)) for m in mt.meta]
    sp = mt.split_ms()
    kernels = {
        "linear_h3p": {"bound": "mfma", "launches_per_step": len(hd) // 3, "ms_per_step": round(sum(hd) / 3, 3),
                       "achieved": round(sum(flops) / (sum(hd) * 1e-3) / 1e12, 1) if hd else 0.0, "peak": MFMA_F16_PEAK_TFLOPS,
                       "unit": "TFLOP/s of fp16 products", "frac": round(sum(flops) / (sum(hd) * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4) if hd else 0.0,
                       "k_le_320": {"launches_per_step": len(k300) // 3, "ms_per_step": round(sum(hd[i] for i in k300) / 3, 3),
                                    "achieved": round(sum(flops[i] for i in k300) / max(sum(hd[i] for i in k300), 1e-9) / 1e9, 1)}},
        # (since round 5 the C = 300 layers run the edge-logits pair: the bracket holds isg_gatv2_edge_logits -- lin_edge inside --
        # AND the flat kernel from logits; bytes_mp still counts e_proj, which neither writes nor reads: a lower bound, as at C = 128)
        "gatv2_mp_pair": {"bound": "hbm", "what": "isg_gatv2_edge_logits (rows kernel, lin_edge inside) + isg_gatv2_mp_fwd_logits_planes "
                                                  "(flat kernel from logits) against SURVEY 8(d)'s bytes_mp at H C = 1200",
                          "launches_per_step": len(md) // 3, "ms_per_step": round(sum(md) / 3, 3),
                          "parts_us": [round(sum(p[i] for p in sp) / max(len(sp), 1) * 1e3, 1) for i in (0, 1)] if sp else None,
                          "achieved": round(sum(mpb) / max(sum(md), 1e-9) / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                          "frac": round(sum(mpb) / max(sum(md), 1e-9) / 1e6 / HBM_PEAK_GBPS, 4),
                          "algorithmic_bytes_per_launch": int(sum(mpb) / max(len(mpb), 1))},
    }
    return {"kernels": kernels,
            "workload": "BASELINE configs[2] stand-in: full ISubGVQA model, C=300, 4 MGAT layers, I-MLE k=5, 12-token "
                        "questions, GQA-shaped synthetic scene graphs (no GQA data in the container)",
            "graphs": graphs, "nodes": int(wl.x.size(0)), "edges": int(wl.edge_index.size(1)),
            "ms_per_step": round(dt * 1e3, 3), "questions_per_s": round(graphs / dt, 1), "steps": steps,
            # isg_linear_h3p's large-result store policy as measured on this box in the warm-up (ops._h3p_tune): -1 = the
            # library's choice, 2 = write-through streaming; `us`: the two medians it compared
            "h3p_store_policy": ops.h3p_store_policy()}


TRAFFIC_SOURCES = {      # kind of summary -> the kernel sources it measured (csrc/); a summary is only as good as their bytes
    "layer_conv": ("isg_layer_conv.hip",), "tile_conv": ("isg_layer_tile.hip",), "graph": ("isg_mp_graph.hip",),
    "chunk": ("isg_mp.hip",), "logits_pair": ("isg_mp_logits.hip", "isg_mp_graph.hip"),
}


def kernel_source_hash(kind: str, root: str = None) -> str:
    """sha256 over the source files of the kernel(s) a traffic summary of this kind describes."""
    import hashlib
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES[kind]:
        h.update(open(os.path.join(root or ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def load_traffic(N: int, E: int, kernel: str, with_source: bool = False, profiles_dir: str = None):
    """HBM bytes per message-passing launch from the committed PMC summary (profiles/*_mp_traffic.json, made by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), if kernel, batch shape AND kernel
    source match this run; None otherwise.  kernel: "layer_conv" | "tile_conv" | "graph" | "chunk" | "logits_pair".  A summary
    carries the sha256 of the kernel source(s) it was measured on (`source_sha256`): one without it, or measured on other
    bytes, is stale and refused -- a kernel edit invalidates the replayed number instead of silently keeping it.
    with_source: return (bytes, file name) / (None, why)."""
    import glob
    why = "no summary for this kernel and batch shape"
    for path in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*mp_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
            name = t.get("kernel", "")
            if "layer_conv" in name:
                kind = "layer_conv"
            elif "tile_conv" in name:
                kind = "tile_conv"
            elif "edge_logits" in name or "+" in name:
                kind = "logits_pair"          # whatever else the file says: two kernels were summed
            else:
                kind = t.get("kind") or ("graph" if "graph" in name else "chunk")
            if t.get("N") == N and t.get("E") == E and kind == kernel:
                if t.get("source_sha256") != kernel_source_hash(kind):
                    why = f"{os.path.basename(path)} was measured on another version of {' + '.join(TRAFFIC_SOURCES[kind])}"
                    continue
                return (t.get("hbm_bytes_per_launch"), os.path.basename(path)) if with_source else t.get("hbm_bytes_per_launch")
        except Exception:
            pass
    return (None, why) if with_source else None


def time_unfused_mp(wl, cfg, dev, launches: int = 20):
    """The un-fused message-passing kernel (isg_gatv2_mp_fwd_rowmax, e_proj streamed) on this run's batch: 3 warm + `launches`
    timed launches with HIP events, a 512 MiB write between them (cold caches, as between the layers of a step).  NOT part of
    the timed step: the step runs the edge-logits pair; this keeps the round-1 roofline figure measurable."""
    import torch
    from isubgvqa_amd import ops
    N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
    plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
    g = torch.Generator(device=dev).manual_seed(7)
    x_lr = torch.randn(N, 2 * H * C, device=dev, generator=g)
    e_proj = torch.randn(E, H * C, device=dev, generator=g)
    att = torch.randn(1, H, C, device=dev, generator=g)
    bias = torch.randn(H * C, device=dev, generator=g)
    flush = torch.empty(1 << 27, device=dev)
    ts = []
    for r in range(launches + 3):
        flush.fill_(float(r))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gatv2_mp(x_lr[:, :H * C], x_lr[:, H * C:], e_proj, att, plan, H, bias=bias, want_rowmax=True)
        e.record()
        torch.cuda.synchronize()
        if r >= 3:
            ts.append(s.elapsed_time(e))
    ms = sum(ts) / len(ts)
    b = ops.mp_algorithmic_bytes(N, E, H, C, False)
    gbps = b / (ms * 1e-3) / 1e9
    return {"kernel": "gatv2_mp_graph_kernel<2,1> (isg_gatv2_mp_fwd_rowmax, e_proj streamed) -- not in the timed step",
            "avg_launch_us": round(ms * 1e3, 2), "algorithmic_bytes_per_launch": int(b), "achieved": round(gbps, 1),
            "frac": round(gbps / HBM_PEAK_GBPS, 4), "frac_of_measured_copy": round(gbps / HBM_COPY_GBPS, 4),
            "traffic": load_traffic(N, E, "graph"), "launches_timed": len(ts)}


# keys of the ONE JSON line (N = 1, default flags); validate_line() runs before the line is printed and in a CPU test over the
# committed sample (profiles/r03_*bench.json)
LINE_SCHEMA = {
    "": ["metric", "value", "unit", "summary", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
         "dtype", "data", "config", "roofline", "cpu_baseline", "rccl", "fallbacks", "dense_err_vs_fp32", "cfg5", "full_model", "mixed",
         "sustained"],
    "config": ["workload", "graphs_per_gpu", "global_batch", "parallelism"],
    "roofline": ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us"],
    "cpu_baseline": ["value", "unit", "cores", "kind", "sample", "cfg1"],
    "fallbacks": ["torch_linear", "torch_layer_norm", "torch_attention", "row_absmax"],
    "dense_err_vs_fp32": ["max"],
    "cfg5": ["workload", "graphs", "ms_per_step", "questions_per_s", "mp_kernel", "mp_avg_launch_us", "mp_achieved_GBps",
             "mp_frac_of_hbm_peak", "imbalance_world8_max_over_mean", "fp32_rows"],
    "full_model": ["workload", "graphs", "ms_per_step", "questions_per_s", "h3p_store_policy", "kernels"],
    "sustained": ["seconds", "steps", "ms_per_step", "ratio_to_burst"],
    "mixed": ["workload", "graphs", "big_graphs", "nodes", "ms_per_step", "host_issue_ms_per_step", "dispatch",
              "tile_kernel_node_share", "per_graph_kernels_ms_per_step"],
}


def validate_line(res: dict, full: bool = True) -> None:
    """Raise if the line lacks a key the contract or VERDICT asks for.  full=False: a run with legs switched off (--no-*)."""
    for section, keys in LINE_SCHEMA.items():
        obj = res if section == "" else res.get(section)
        if obj is None:
            if full and section != "":
                raise KeyError(f"bench line lacks the {section!r} object")
            continue
        missing = [k for k in keys if k not in obj and (full or section != "")]
        if missing:
            raise KeyError(f"bench line{' / ' + section if section else ''} lacks {missing}")
    if full and set(res["cfg5"]["imbalance_world8_max_over_mean"]) != {"equal_counts", "balanced"}:
        raise KeyError("cfg5.imbalance_world8_max_over_mean needs equal_counts and balanced")


def dense_err_vs_fp32(dev):
    """How close to fp32 the exact-split dense kernels of the step are: for each of them, max |kernel - fp64| / max |torch fp32
    GEMM - fp64| on the same operands (1.0 = as accurate as a plain fp32 GEMM; the dtype field says "f32" on that basis).
    Computed once, outside the timed region, at the step's K and N with 16 384 rows."""
    import torch
    from isubgvqa_amd import ops, synthetic
    g = torch.Generator(device=dev).manual_seed(3)
    out = {}

    def ratio(got, a, w, b=None, act=None):
        ref = a.double() @ w.double().t()
        base = a @ w.t()
        if b is not None:
            ref, base = ref + b.double(), base + b
        if act is not None:
            ref, base = act(ref), act(base)
        return round(((got.double() - ref).abs().max() / (base.double() - ref).abs().max().clamp_min(1e-30)).item(), 3)

    M = 16384
    a = torch.randn(M, 128, device=dev, generator=g) * torch.rand(M, 1, device=dev, generator=g).mul(6).exp()
    w = torch.randn(1024, 128, device=dev, generator=g) * 0.1
    b = torch.randn(1024, device=dev, generator=g)
    out["lin_l|lin_r (isg_linear_f16x3, 128 -> 1024)"] = ratio(ops.linear(a, w, b), a, w, b)
    w = torch.randn(512, 384, device=dev, generator=g) * 0.05
    a3 = torch.randn(4096, 384, device=dev, generator=g)
    out["embedding (384 -> 512, GELU)"] = ratio(ops.linear(a3, w, None, gelu=True), a3, w, None, torch.nn.functional.gelu)
    # the fused dense tail: x_proj.0 -> GELU -> x_proj.2 -> GELU of isg_mgat_dense_tail, read back through an identity tail
    cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": 512})
    wl = synthetic.make_workload(cfg).to(dev)
    net = synthetic.build_answer_model(cfg).to(dev).eval()
    xp = net.gat_seq.x_proj[0]
    N = wl.x.size(0)
    plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes, max_edges=wl.max_edges)
    co = torch.randn(N, 512, device=dev, generator=g) * torch.rand(N, 1, device=dev, generator=g).mul(4).exp()
    c64 = torch.nn.functional.gelu(torch.nn.functional.linear(
        torch.nn.functional.gelu(torch.nn.functional.linear(co.double(), xp[0].weight.double(), xp[0].bias.double())),
        xp[2].weight.double(), xp[2].bias.double()))
    c32 = xp(co)
    if ops.dense_tail_supported(plan, xp, 512, 128):
        # one-node "graphs" would be needed to read c back exactly; instead compare the whole fused layer tail with the same tail
        # applied (by this library's un-fused kernel, whose arithmetic the fused kernel repeats) to the fp64 / fp32 c
        ops.attach_row_maxima(co, co.view(N, 4, 128).abs().amax(dim=2).contiguous())
        bn = net.gat_seq.bns[0]
        ins, h = wl.instr[0].contiguous(), torch.zeros(N, 128, device=dev)
        got = ops.mgat_dense_tail(co, xp, ins, h, plan, bn.weight, bn.bias, bn.mean_scale, bn.eps)[0]
        tail = lambda c: ops.mgat_layer_tail(ins, c.float().contiguous(), h, plan, bn.weight, bn.bias, bn.mean_scale, bn.eps)
        ref, base = tail(c64), tail(c32)
        out["x_proj pair inside isg_mgat_dense_tail (512 -> 256 -> 128, through the layer tail)"] = round(
            ((got - ref).abs().max() / (base - ref).abs().max().clamp_min(1e-30)).item(), 3)
    out["max"] = max(out.values())
    return out


def eval_batch_leg(dev, graphs: int = 1024, steps: int = 200):
    """The reference's evaluation batch (datasets/build.py:59-62: 4 x --batch-size graphs; 1 024 at the configs[1] distribution) on the
    PRODUCT path, eager against its own hipGraph option (`AnswerModel.forward(..., capture=True)` = ops.StepCapture): at this size
    the step is launch-bound, which is what the option is for.  Noise from torch's generator in both (the captured
    step draws it inside the graph)."""
    import torch
    from isubgvqa_amd import synthetic
    cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs, "seed": 4242})
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    res = {}
    with torch.no_grad():
        for name, kw in (("eager", {}), ("captured", {"capture": True})):
            for _ in range(5):
                model(wl, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = model(wl, **kw)[0]
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / steps
        assert torch.isfinite(out).all()
        model._step_capture.verify()
    return {"workload": "the reference's evaluation batch: configs[1] distribution, in-model Gumbel noise from torch's generator",
            "graphs": graphs, "nodes": int(wl.x.size(0)), "steps": steps,
            "eager_ms_per_step": round(res["eager"] * 1e3, 4), "captured_ms_per_step": round(res["captured"] * 1e3, 4),
            "captured_questions_per_s": round(graphs / res["captured"], 1),
            "option": "AnswerModel.forward(..., capture=True) / ISubGVQA.forward(..., capture=True): ops.StepCapture, one hipGraph per batch shape"}


def small_batch_leg(dev, sizes=(1, 8, 32), steps: int = 100):
    """The FULL model (C = 300, 4 MGAT layers, question encoder / decoder, scene-graph encoder) at the batch sizes of the reference's own
    evaluation script (run_token_coo.py:49-79: one question per forward): ms per forward, eager, as a replayed hipGraph
    (`ISubGVQA.forward(..., capture=True)`) and with the question side alone replayed (`capture="language"`).  The latency-bound regime: round 6's isg_linear_skinny and the small-batch dispatch."""
    import torch
    from isubgvqa_amd import ops, synthetic
    from isubgvqa_amd.models import build_model
    torch.manual_seed(0)
    model = build_model(synthetic.full_model_args(), None).to(dev).eval()
    out = {"workload": "full ISubGVQA model (configs[2] stand-in) at 1 / 8 / 32 questions of 12 tokens", "steps": steps, "sizes": {}}
    with torch.no_grad():
        for g in sizes:
            wl = synthetic.make_full_workload(g).to(dev)
            sg = wl.scene_graphs()
            res = {}
            # "language": only the question side replayed (keyed by the questions' shape: what a loop over single questions whose
            # scene graphs never repeat a shape can use), the graph side eager
            for name, kw in (("eager", {}), ("captured", {"capture": True}), ("language_captured", {"capture": "language"})):
                f = lambda: model(wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.questions, wl.att_mask, return_masks=True,
                                  scene_graphs=sg, **kw)[0]
                for _ in range(5):
                    f()
                torch.cuda.synchronize()
                ops.reset_counters()
                t0 = time.perf_counter()
                for _ in range(steps):
                    logits = f()
                torch.cuda.synchronize()
                res[name + "_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
                if name == "eager":
                    res["skinny_linears_per_forward"] = round(ops.counters()["linear_skinny"] / steps, 1)
            assert torch.isfinite(logits).all()
            out["sizes"][str(g)] = res
        model._step_capture.verify()
    return out


def mixed_leg(dev, graphs: int = 4096, big: int = 8, steps: int = 10):
    """The configs[1] batch with `big` of its graphs replaced by graphs of 100-189 nodes, after the timed region: the default
    dispatch (tile kernels for the graphs that fit a 64-node / 256-in-edge tile, the others as a batch of their own through the whole
    model on the per-graph kernels and a second stream: ops.run_split) against the per-graph kernels for the whole batch."""
    import torch
    from isubgvqa_amd import ops, synthetic
    gen = torch.Generator().manual_seed(3)
    sizes = synthetic.graph_sizes(synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs}), gen).tolist()
    for i in range(big):
        sizes[(i * 977 + 13) % graphs] = 100 + (i * 37) % 90
    cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": graphs, "sizes": tuple(sizes)})
    wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    out = {}
    keep = ops.MIXED_DISPATCH
    try:
        for name, on in (("mixed", True), ("per_graph", False)):
            ops.MIXED_DISPATCH = on
            with torch.no_grad():
                for i in range(3):
                    model(wl, seed=50 + i)
                torch.cuda.synchronize()
                ops.reset_counters()
                t0 = time.perf_counter()
                for i in range(steps):
                    logits = model(wl, seed=60 + i)[0]
                t_issue = (time.perf_counter() - t0) / steps
                torch.cuda.synchronize()
                out[name] = ((time.perf_counter() - t0) / steps, t_issue, ops.counters())
            assert torch.isfinite(logits).all()
    finally:
        ops.MIXED_DISPATCH = keep
    dt, t_issue, c = out["mixed"]
    visits = c["tile_nodes"] + c["oversize_nodes"]
    return {"workload": f"BASELINE configs[1] with {big} of its graphs replaced by graphs of 100-189 nodes (beyond a graph tile)",
            "graphs": graphs, "big_graphs": big, "nodes": int(wl.x.size(0)), "max_nodes": int(wl.max_nodes),
            "ms_per_step": round(dt * 1e3, 3), "questions_per_s": round(graphs / dt, 1),
            "host_issue_ms_per_step": round(t_issue * 1e3, 3),
            "dispatch": ("tile kernels + the big graphs as a sub-batch on the per-graph kernels (ops.run_split)" if c["oversize_nodes"]
                         else ("tile kernels" if visits else "per-graph kernels")),
            "tile_kernel_node_share": round(c["tile_nodes"] / visits, 4) if visits else 0.0,
            "per_graph_kernels_ms_per_step": round(out["per_graph"][0] * 1e3, 3), "steps": steps}


def cfg5_leg(dev, graphs: int = 2048, steps: int = 10):
    """BASELINE configs[4] on one GPU, after the timed region: skewed graphs (8-200 nodes, power-law in-degree), AIMLE k = 5,
    fp16 feature rows: ms per step, the message-passing kernel against s = 2 bytes_mp, which kernel ran, and how well
    contiguous graph ranges balance sum(nodes + edges) over 8 ranks (host-side arithmetic of distributed.graph_ranges)."""
    import torch
    from isubgvqa_amd import distributed, ops, synthetic
    cfg = synthetic.WorkloadConfig(**{**synthetic.CFG5.__dict__, "num_graphs": graphs, "feature_dtype": "fp16"})
    wl_cpu = synthetic.make_workload(cfg)
    wl = wl_cpu.to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    npg = torch.bincount(wl_cpu.batch, minlength=graphs)
    epg = torch.bincount(wl_cpu.batch[wl_cpu.edge_index[1]], minlength=graphs)
    cost = (npg + epg).double()
    imb = {}
    for bal in (False, True):
        per = [float(cost[lo:hi].sum()) for lo, hi in distributed.graph_ranges(npg, epg, 8, balance=bal)]
        imb["balanced" if bal else "equal_counts"] = round(max(per) / (sum(per) / len(per)), 4)
    deg = torch.bincount(wl_cpu.edge_index[1], minlength=wl_cpu.x.size(0))
    with torch.no_grad():
        for i in range(3):
            model(wl, seed=50 + i)
        torch.cuda.synchronize()
        ops.MP_TIMER = ops.KernelTimer()
        t0 = time.perf_counter()
        for i in range(steps):
            out = model(wl, seed=60 + i)[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        timer, ops.MP_TIMER = ops.MP_TIMER, None
    assert torch.isfinite(out).all()
    # the same batch with fp32 feature rows: mixed dispatch (graphs within a 64-node / 256-slot tile on the tile kernels, the
    # tail of the size distribution on the per-graph kernels: ops.GraphPlan.tile_mode)
    cfg32 = synthetic.WorkloadConfig(**{**synthetic.CFG5.__dict__, "num_graphs": graphs})
    model32 = synthetic.build_answer_model(cfg32).to(dev).eval()
    with torch.no_grad():
        for i in range(3):
            model32(wl, seed=50 + i)
        torch.cuda.synchronize()
        keep = dict(ops.COUNTERS)
        ops.reset_counters()
        t0 = time.perf_counter()
        for i in range(steps):
            out32 = model32(wl, seed=60 + i)[0]
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / steps
        c32 = ops.counters()
        # ... and with the mixed mode forced (its profitability gate open): what the tile kernels + per-graph kernels cost here
        gate = ops.MIXED_MAX_FRACTION, ops.MIXED_MIN_NODES
        ops.MIXED_MAX_FRACTION, ops.MIXED_MIN_NODES = 0.9, 0
        try:
            for i in range(2):
                model32(wl, seed=50 + i)
            torch.cuda.synchronize()
            ops.reset_counters()
            t0 = time.perf_counter()
            for i in range(steps):
                model32(wl, seed=60 + i)
            torch.cuda.synchronize()
            dt32m = (time.perf_counter() - t0) / steps
            c32m = ops.counters()
        finally:
            ops.MIXED_MAX_FRACTION, ops.MIXED_MIN_NODES = gate
        ops.COUNTERS.update(keep)
    assert torch.isfinite(out32).all()
    visits = c32["tile_nodes"] + c32["oversize_nodes"]
    fp32_rows = {"ms_per_step": round(dt32 * 1e3, 3), "questions_per_s": round(graphs / dt32, 1),
                 "dispatch": "mixed: tile kernels + per-graph kernels on the graphs beyond a tile" if c32["oversize_nodes"] else
                             ("tile kernels" if visits else "per-graph kernels (tiles do not pay for this batch)"),
                 "tile_kernel_node_share": round(c32["tile_nodes"] / visits, 4) if visits else 0.0,
                 "mixed_forced": {"ms_per_step": round(dt32m * 1e3, 3),
                                  "tile_kernel_node_share": round(c32m["tile_nodes"] / max(c32m["tile_nodes"] + c32m["oversize_nodes"], 1), 4)}}
    del model32
    durs = timer.durations_ms()
    byt = [ops.mp_algorithmic_bytes(m["N"], m["E"], m["H"], m["C"], m["masked"], m.get("feat_bytes", 4)) for m in timer.meta]
    mp_ms = sum(durs) / max(len(durs), 1)
    gbps = (sum(byt) / max(len(byt), 1)) / (mp_ms * 1e-3) / 1e9 if durs else 0.0
    small = wl.max_nodes <= 64 and wl.max_edges <= 256
    fused = bool(timer.meta) and all(m.get("fused_logits") for m in timer.meta)
    parts = timer.split_ms()
    return {"workload": "BASELINE configs[4] on one GPU: skewed graphs (8-200 nodes, Pareto sizes, power-law in-degree), AIMLE k=5, "
                        "fp16 feature rows / fp32 arithmetic, 3 layers C=128 H=4",
            "graphs": graphs, "nodes": int(wl.x.size(0)), "edges": int(wl.edge_index.size(1)), "max_nodes": int(wl.max_nodes),
            "max_edges": int(wl.max_edges), "max_in_degree": int(deg.max()), "ms_per_step": round(dt * 1e3, 3),
            "questions_per_s": round(graphs / dt, 1),
            # since round 5 (ABI v19) the layers run the pair on half rows: ONE bracket holds isg_gatv2_edge_logits_f16 (lin_edge inside,
            # e_proj rounded to half in registers) and isg_gatv2_mp_fwd_logits_f16; bytes_mp (s = 2) still counts e_proj: a lower bound
            "mp_kernel": ("isg_gatv2_edge_logits_f16 (rows kernel) + isg_gatv2_mp_fwd_logits_f16 (gatv2_mp_graph_kernel<.., "
                          + ("64, 256" if small else "256, 1024: per-graph tables for hub graphs") + ", f16>)") if fused else
                         (("gatv2_mp_graph_kernel<.., 64, 256, f16>" if small else "gatv2_mp_graph_kernel<.., 256, 1024, f16> (per-graph "
                           "tables for hub graphs)") + " (isg_gatv2_mp_fwd_f16; e_proj streamed as half rows)"),
            "mp_parts_us": [round(sum(p[i] for p in parts) / len(parts) * 1e3, 1) for i in (0, 1)] if parts else None,
            "mp_avg_launch_us": round(mp_ms * 1e3, 2), "mp_algorithmic_bytes_per_launch_s2": int(sum(byt) / max(len(byt), 1)),
            "mp_achieved_GBps": round(gbps, 1), "mp_frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
            "imbalance_world8_max_over_mean": imb, "steps": steps, "fp32_rows": fp32_rows}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))
    if int(world_env or "1") != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world_env} does not match --gpus {args.gpus}: start `python bench.py --gpus N` "
                         "without a launcher, or give torch.distributed.run --nproc-per-node the same N")
    import torch
    import torch.distributed as dist
    from isubgvqa_amd import ops, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the product path has no CPU fallback"
    # ISG_BENCH_SINGLE_DEVICE=1 + ISG_BENCH_BACKEND=gloo: rehearse the N>1 code path with every rank on cuda:0
    # (a one-GPU box cannot form an RCCL communicator with two ranks on one device)
    if os.environ.get("ISG_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("ISG_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    ops.MP_KERNEL = args.mp_kernel
    ops.GEMM_BACKEND = args.gemm
    ops.FUSE_LOGITS = not args.no_fuse_logits
    from isubgvqa_amd.distributed import GatherPipeline, shard_workload
    cfg5 = args.workload == "cfg5"
    shard_cost = None
    if cfg5:
        # BASELINE configs[4]: every rank generates the SAME batch of world x --graphs skewed graphs on the host (seeded) and keeps
        # its contiguous range, balanced by sum(nodes + edges); the ranks' graph counts differ
        import dataclasses
        cfg_all = synthetic.WorkloadConfig(**{**synthetic.CFG5.__dict__, "num_graphs": world * args.graphs,
                                              "feature_dtype": args.features})
        wl_host = shard_workload(synthetic.make_workload(cfg_all), rank, world, balance=True)
        cfg = dataclasses.replace(cfg_all, num_graphs=wl_host.num_graphs)
        shard_cost = int(wl_host.x.size(0) + wl_host.edge_index.size(1))
        wl = wl_host.to(dev)
        del wl_host
    else:
        cfg = synthetic.WorkloadConfig(**{**synthetic.CFG2.__dict__, "num_graphs": args.graphs,
                                          "seed": synthetic.CFG2.seed + rank, "feature_dtype": args.features})
        wl = synthetic.make_workload(cfg).to(dev)
    model = synthetic.build_answer_model(cfg).to(dev).eval()
    N, E = wl.x.size(0), wl.edge_index.size(1)
    total_graphs = world * args.graphs                  # cfg2: world x B_local; cfg5: the one batch all ranks share
    # the per-step collective: up to --gather-depth all-gathers in flight on the communicator's stream beside the next steps' kernels
    pipe = GatherPipeline(cfg.num_graphs, 1842, dev, what=args.gather, depth=args.gather_depth, ragged=cfg5)
    drain = pipe.drain

    def step(i: int):
        logits, mask, gate = model(wl, seed=1000 + i, use_hints=not args.no_hints)   # in-kernel Philox noise
        return pipe.submit(i, logits)

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graph = None
    t_issue = t_local = None
    step_counters = None
    if args.launch == "graph":
        if world > 1:
            raise SystemExit("--launch graph: single GPU (the all-gather stays outside a captured step)")
        from isubgvqa_amd import synthetic as _syn
        noise_layers = [i for i, t in enumerate(cfg.masks) if t != 1.0]
        cap = {}

        def body():
            cap["plan"] = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs,
                                              max_nodes=wl.max_nodes, max_edges=wl.max_edges)
            noises = {i: _syn.gumbel_noise((cfg.num_graphs, wl.max_nodes), dev) for i in noise_layers}
            return model(wl, noises=noises, plan=cap["plan"])[0]

    with torch.no_grad():
        if args.launch == "graph":
            for i in range(max(args.warmup, 2)):      # eager: kernel attributes, weight planes, allocator pools, hint check
                out = body()
            ops.check_plans()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = body()
            graph.replay()
            fence()
            progress("step captured as a hipGraph")
            t0 = time.perf_counter()
            for i in range(args.steps):
                graph.replay()
            fence()
            dt = time.perf_counter() - t0
            cap["plan"].verify_hints()
            # kernel durations for the roofline: EAGER steps after the timed region (events cannot sit inside a graph)
            ops.reset_counters()
            ops.MP_TIMER = ops.KernelTimer()
            for i in range(10):
                step(i)
            fence()
            timer, ops.MP_TIMER = ops.MP_TIMER, None
            # the captured step launches what these eager steps launch: their counters are the step's
            step_counters = {k: round(v / 10, 3) for k, v in ops.counters().items()}
        else:
            for i in range(args.warmup):
                step(i)
            fence()
            ops.reset_counters()
            ops.MP_TIMER = ops.KernelTimer()
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = step(args.warmup + i)
            t_issue = time.perf_counter() - t0      # every step launched; the last all-gather is still in flight
            drain()
            torch.cuda.synchronize()
            t_local = time.perf_counter() - t0      # this rank's kernels and its last collective done (before the barrier)
            fence()
            dt = time.perf_counter() - t0
            timer, ops.MP_TIMER = ops.MP_TIMER, None
            step_counters = {k: round(v / max(args.steps, 1), 3) for k, v in ops.counters().items()}
    assert torch.isfinite(out).all()

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_own = dt
    dt = float(t.item())
    # where an N > 1 run loses time: every rank's own step time (before the closing barrier) and how long it then waited
    per_rank = None
    if world > 1 and t_local is not None:
        mine = torch.tensor([t_local / args.steps * 1e3, (dt_own - t_local) * 1e3, t_issue / args.steps * 1e3],
                            dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(v[0]), 4) for v in allr],
                    "closing_barrier_wait_ms": [round(float(v[1]), 3) for v in allr],
                    "host_issue_ms_per_step": [round(float(v[2]), 4) for v in allr]}
        if cfg5:       # what the balanced partition gave every rank: graphs, sum(nodes + edges), and the imbalance of the latter
            mine = torch.tensor([cfg.num_graphs, shard_cost], dtype=torch.int64, device=dev)
            allc = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allc, mine)
            costs = [int(v[1]) for v in allc]
            per_rank.update(graphs=[int(v[0]) for v in allc], nodes_plus_edges=costs,
                            imbalance_max_over_mean=round(max(costs) / (sum(costs) / len(costs)), 4))
    # evidence that the communicator saw every rank: backend, world size, each rank's device index and GPU name
    devs = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=dev)
    if world > 1:
        all_devs = [torch.empty_like(devs) for _ in range(world)]
        dist.all_gather(all_devs, devs)
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(),
                "devices": [int(d.item()) for d in all_devs], **pipe.describe()}
    else:
        rccl = {"backend": None, "world": 1, "devices": [int(devs.item())], "collective": None}
    rccl["gpu"] = torch.cuda.get_device_name(dev)
    props = torch.cuda.get_device_properties(dev)       # the marketing string is generic on this image; the ISA name is not
    rccl["gcn_arch"] = getattr(props, "gcnArchName", None)
    rccl["compute_units"] = int(props.multi_processor_count)

    durs = timer.durations_ms()
    bytes_l = [ops.mp_algorithmic_bytes(m["N"], m["E"], m["H"], m["C"], m["masked"], m.get("feat_bytes", 4)) for m in timer.meta]
    mp_ms = sum(durs) / max(len(durs), 1)
    mp_bytes = sum(bytes_l) / max(len(bytes_l), 1)
    achieved = mp_bytes / (mp_ms * 1e-3) / 1e9 if durs else 0.0
    # The reference's message + aggregate runs as TWO launches when lin_edge is folded into the logits
    # (isg_gatv2_edge_logits + isg_gatv2_mp_fwd_logits): the bracket then covers both -- it also contains the lin_edge GEMM,
    # which the un-fused kernel's bracket did not -- and `achieved` still divides SURVEY 8(d)'s bytes_mp (e_proj included)
    # by it: a lower bound on what the round-1 definition would give, not comparable with a bracket of the MP kernel alone.
    fused = bool(timer.meta) and all(m.get("fused_logits") for m in timer.meta)
    tile_conv = bool(timer.meta) and all(m.get("tile_conv") for m in timer.meta)
    layer_conv = tile_conv and all(m.get("layer_conv") for m in timer.meta)
    parts = None
    if fused:
        split = timer.split_ms()
        own = [(ops.edge_logits_algorithmic_bytes(m["N"], m["E"], m["H"], m["C"], m["K"], m["masked"], feat_bytes=m.get("feat_bytes", 4)),
                ops.mp_logits_algorithmic_bytes(m["N"], m["E"], m["H"], m["C"], m["masked"], feat_bytes=m.get("feat_bytes", 4)))
               for m in timer.meta]
        n = max(len(split), 1)
        t0, t1 = sum(a for a, _ in split) / n, sum(b for _, b in split) / n
        b0, b1 = sum(a for a, _ in own) / n, sum(b for _, b in own) / n
        parts = [{"kernel": "gatv2_edge_logits_kernel (isg_gatv2_edge_logits: lin_edge GEMM + row gathers -> logits[E,H])",
                  "avg_launch_us": round(t0 * 1e3, 2), "own_algorithmic_bytes": int(b0),
                  "own_achieved_GBps": round(b0 / (t0 * 1e-3) / 1e9, 1), "own_frac": round(b0 / (t0 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                 {"kernel": "gatv2_mp_graph_kernel<2,1> (isg_gatv2_mp_fwd_logits: softmax + aggregation from logits)",
                  "avg_launch_us": round(t1 * 1e3, 2), "own_algorithmic_bytes": int(b1),
                  "own_achieved_GBps": round(b1 / (t1 * 1e-3) / 1e9, 1), "own_frac": round(b1 / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}]
    # continuity with round 1: the UN-fused message-passing kernel on the same batch, timed here outside the timed region
    unfused = None
    if (fused or tile_conv) and rank == 0 and not cfg5:
        unfused = time_unfused_mp(wl, cfg, dev)

    if rank == 0:
        traffic, traffic_src = load_traffic(N, E, "layer_conv" if layer_conv else ("tile_conv" if tile_conv else
                                                                                   ("logits_pair" if fused else args.mp_kernel)),
                                            with_source=True)
        res = {
            "metric": "GQA questions/sec", "value": round(total_graphs * args.steps / dt, 1),
            "unit": "questions/s",
            # the other legs' headline numbers, flat and at the FRONT of the line (filled in below as the legs run): a record that
            # keeps only the head of the line keeps them
            "summary": {"configs1_ms_per_step": round(dt / args.steps * 1e3, 4) if not cfg5 else None},
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.features == "fp32" else "f32 arithmetic on f16 feature rows", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: ONE batch of skewed synthetic scene graphs (8-200 nodes, Pareto sizes, power-law "
                                    "in-degree) sharded into contiguous graph ranges balanced by sum(nodes + edges); MGAT(3 masked-GATv2 "
                                    "layers, C=128, H=4, masks [1,1,0.15], AIMLE k=5, fp16 feature rows / fp32 arithmetic) + "
                                    "GlobalAttention pooling + 1842-way classifier; graph plan (CSR) built every step; ragged logits "
                                    "all-gather") if cfg5 else
                                   "BASELINE configs[1]: MGAT(3 masked-GATv2 layers, C=128, H=4, masks [1,1,0.15], Gumbel "
                                   "top-k k=5) + GlobalAttention pooling + 1842-way classifier over synthetic "
                                   "GQA-shaped scene graphs (~20 nodes, ~50 edges); graph plan (CSR) built every step; "
                                   "question encoder/decoder not included (full model needs C=300, SURVEY §5.1)",
                       "graphs_per_gpu": args.graphs if cfg5 else cfg.num_graphs, "global_batch": total_graphs,
                       "nodes_per_gpu": N, "edges_per_gpu": E, "channels": cfg.channels, "heads": cfg.heads,
                       "layers": cfg.layers, "sampler": ("aimle" if cfg5 else "gumbel") + "(in-kernel Philox noise)", "k": cfg.sample_k,
                       "parallelism": (f"dp{world} (graphs sharded, RCCL all-gather of logits" + (", ragged shards" if cfg5 else "") + ")") if world > 1 else "dp1",
                       "feature_rows": args.features,
                       "launch": "eager" if graph is None else "hipgraph: one captured step (plan build + model) replayed; Gumbel noise from torch's generator inside the graph; the roofline's kernel durations from eager steps after the timed region", "edge_projection": "unfused" if args.no_fuse_logits else ("inside isg_gatv2_layer_conv (with lin_l | lin_r)" if layer_conv else "inside isg_gatv2_tile_conv" if tile_conv else "folded into the logits"), "layer_tail": "isg_mgat_dense_tail (x_proj + instruction attention + GraphNorm + residual + next gate, one launch per layer)" if ops.FUSE_DENSE_TAIL else "un-fused", "dense": ("exact-split fp32 Linears on MFMA: isg_linear_f16x3 / _f16x3_tile (2 fp16 planes, 3 products, per-row scales), isg_linear_bf16x6 for the small ones" if args.gemm == "bf16x6" else "hipBLASLt fp32 via torch")},
            "roofline": {"bound": "hbm",
                         "kernel": ("gatv2_layer_conv_kernel (isg_gatv2_layer_conv: the reference's message + aggregate WITH lin_edge AND "
                                    "lin_l | lin_r inside as ONE persistent launch on graph-aligned tiles)") if layer_conv else
                                   ("gatv2_tile_conv_kernel (isg_gatv2_tile_conv: the reference's message + aggregate WITH lin_edge "
                                    "inside as ONE persistent launch on graph-aligned tiles)") if tile_conv else
                                   ("isg_gatv2_edge_logits + isg_gatv2_mp_fwd_logits (the reference's message + aggregate WITH "
                                    "lin_edge inside: two launches, one bracket)") if fused else
                                   (("gatv2_mp_graph_kernel<2,1>" if args.mp_kernel == "graph" else "gatv2_mp_kernel<4,2>") + " (isg_gatv2_mp_fwd)"),
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "frac_of_measured_copy": round(achieved / HBM_COPY_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(mp_bytes),
                         "avg_launch_us": round(mp_ms * 1e3, 2), "launches_timed": len(durs)},
        }
        res["summary"].update(roofline_frac=res["roofline"]["frac"], mp_avg_launch_us=res["roofline"]["avg_launch_us"])
        if tile_conv:
            m0 = timer.meta[0]
            # own minimum: edge planes + (x_l and x_r, or with lin_l | lin_r inside: the gated node rows) in, out + alpha back, CSR
            own = sum(4 * m["E"] * m["K"] + (4 * m["N"] * m["K_in"] + 4 * m["N"] * m["H"] * m["C"] if m.get("layer_conv") else
                                              12 * m["N"] * m["H"] * m["C"]) + 4 * m["E"] * m["H"] + 16 * m["E"] +
                      (4 * m["E"] if m["masked"] else 0) for m in timer.meta) / len(timer.meta)
            res["roofline"]["note"] = ("bytes_mp of SURVEY 8(d) (x_l, x_r, e_proj included -- this kernel writes or reads none of them) "
                                       "over the time of the ONE launch that also contains the lin_edge GEMM" +
                                       (" and the lin_l | lin_r GEMM" if layer_conv else "") + ": a lower bound on the round-1 "
                                       "definition, the bracket holds strictly more work; own_* = the kernel against its own minimum "
                                       "traffic")
            res["roofline"]["own_algorithmic_bytes"] = int(own)
            res["roofline"]["own_achieved_GBps"] = round(own / (mp_ms * 1e-3) / 1e9, 1)
            res["roofline"]["own_frac"] = round(own / (mp_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            flops = 3 * 2.0 * m0["E"] * m0["H"] * m0["C"] * m0["K"]
            if layer_conv:
                flops += 3 * 2.0 * m0["N"] * 2 * m0["H"] * m0["C"] * m0["K_in"]
            res["roofline"]["mfma_products_TFLOPs"] = round(flops / (mp_ms * 1e-3) / 1e12, 1)
            # `bound` above is the contract's label for the path (graph / segmented-reduce work prices against HBM); what the
            # counters say limits THIS kernel is neither of the two rooflines:
            res["roofline"]["own_bound"] = {
                "limit": "issue + barrier lock-step: eight waves pass ~7 barriers per (tile, head), so vector issue, the matrix "
                         "pipe and LDS add up instead of overlapping",
                "hbm_frac_on_own_bytes": res["roofline"]["own_frac"],
                "mfma_frac_of_dense_fp16_peak": round(flops / (mp_ms * 1e-3) / 1e12 / 2500.0, 4),
                "evidence": "profiles/r05_q_tile_kernels_sq_counters.txt (round 5: matrix cores 31 % busy, LDS 34 %, SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.43), "
                            "profiles/r03_bg_layer_conv_ablation.md (the phases add)"}
            if layer_conv:     # the launch also IS lin_l | lin_r: the bytes the reference's projection moves beside bytes_mp
                proj = sum(4 * m["N"] * m["K_in"] + 8 * m["N"] * m["H"] * m["C"] for m in timer.meta) / len(timer.meta)
                res["roofline"]["with_projection_bytes"] = {
                    "what": "bytes_mp + the lin_l | lin_r projection's own traffic (x in, x_l and x_r out), the two reference steps "
                            "this one launch performs; `frac` above counts bytes_mp only",
                    "algorithmic_bytes": int(mp_bytes + proj),
                    "achieved": round((mp_bytes + proj) / (mp_ms * 1e-3) / 1e9, 1),
                    "frac": round((mp_bytes + proj) / (mp_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            res["roofline"]["unfused_kernel"] = unfused
        if fused:
            res["roofline"]["note"] = ("bytes_mp of SURVEY 8(d) (e_proj included, which this pair never writes or reads) over the "
                                       "time of BOTH launches, lin_edge GEMM included: a lower bound, not comparable with the "
                                       "round-1 bracket of the message-passing kernel alone; see parts and unfused_kernel")
            res["roofline"]["parts"] = parts
            res["roofline"]["unfused_kernel"] = unfused
        res["rccl"] = rccl
        if per_rank is not None:
            res["per_rank"] = per_rank
        # launches per step that left this library's dense kernels (hipBLASLt / torch LayerNorm / torch attention) and extra
        # row-maximum passes: DESIGN section 1's "no GEMM of the inference path runs on hipBLASLt" as a number
        res["fallbacks"] = step_counters
        if world == 1:
            progress("dense_err_vs_fp32 leg")
            with torch.no_grad():
                res["dense_err_vs_fp32"] = dense_err_vs_fp32(dev)
        if world == 1 and not cfg5 and not args.no_sustained:
            # >= 2 s of back-to-back configs[1] steps: the clocks under load, and something the driver's busy sampler can see
            progress("sustained leg (>= 2 s of steps)")
            with torch.no_grad():
                n_s = max(200, int(args.sustained_seconds / max(dt / args.steps, 1e-6)))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n_s):
                    step(args.warmup + args.steps + i)
                torch.cuda.synchronize()
                ds = time.perf_counter() - t0
            res["sustained"] = {"seconds": round(ds, 3), "steps": n_s, "ms_per_step": round(ds / n_s * 1e3, 4),
                                "ratio_to_burst": round((ds / n_s) / (dt / args.steps), 4)}
            res["summary"]["sustained_ms_per_step"] = res["sustained"]["ms_per_step"]
        if world == 1 and not cfg5 and not args.no_full_model:
            del model, wl
            torch.cuda.empty_cache()
            progress(f"configs[1] step timed: {dt / args.steps * 1e3:.3f} ms; full model leg ({args.full_model_graphs} graphs)")
            res["full_model"] = full_model_rate(dev, args.full_model_graphs)
            fk = res["full_model"]["kernels"]
            res["summary"].update(full_model_ms_per_step=res["full_model"]["ms_per_step"],
                                  full_model_questions_per_s=res["full_model"]["questions_per_s"],
                                  full_model_linear_h3p_frac_of_mfma_peak=fk["linear_h3p"]["frac"],
                                  full_model_mp_pair_frac_of_hbm_peak=fk["gatv2_mp_pair"]["frac"])
        if world == 1 and not cfg5 and not args.no_cfg5:
            progress("cfg5 leg (skewed graphs, AIMLE, fp16 rows)")
            res["cfg5"] = cfg5_leg(dev)
            res["summary"].update(cfg5_ms_per_step=res["cfg5"]["ms_per_step"], cfg5_mp_frac_of_hbm_peak=res["cfg5"]["mp_frac_of_hbm_peak"])
        if world == 1 and not cfg5 and not args.no_mixed:
            progress("mixed leg (configs[1] + 8 graphs beyond a tile)")
            res["mixed"] = mixed_leg(dev)
            res["summary"].update(mixed_ms_per_step=res["mixed"]["ms_per_step"],
                                  mixed_host_issue_ms_per_step=res["mixed"]["host_issue_ms_per_step"])
        if world == 1 and not cfg5 and not args.no_mixed:
            progress("eval_batch leg (1024 graphs: eager vs the product path's hipGraph option)")
            res["eval_batch"] = eval_batch_leg(dev)
            res["summary"].update(eval_batch_1024_eager_ms=res["eval_batch"]["eager_ms_per_step"],
                                  eval_batch_1024_captured_ms=res["eval_batch"]["captured_ms_per_step"])
        if world == 1 and not cfg5 and not args.no_full_model:
            progress("small_batch leg (full model at 1 / 8 / 32 questions: eager vs capture=True)")
            res["small_batch"] = small_batch_leg(dev)
            sb = res["small_batch"]["sizes"]
            res["summary"].update(full_model_1_question_eager_ms=sb["1"]["eager_ms"], full_model_1_question_captured_ms=sb["1"]["captured_ms"],
                                  full_model_1_question_language_captured_ms=sb["1"]["language_captured_ms"],
                                  full_model_8_questions_captured_ms=sb["8"]["captured_ms"])
        ops.check_plans()           # any understated GraphPlan hint of this run raises here
        if world == 1 and not cfg5 and not args.no_cpu_baseline:
            progress("cpu_baseline leg (oracle on the host cores)")
            res["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample_graphs, args.cpu_seconds, args.cpu_threads)
            res["cpu_baseline"]["cpu"] = cpu_model_string()
            res["cpu_baseline"]["cfg1"] = cpu_baseline_cfg1()
            res["cpu_baseline"]["single_question"] = cpu_baseline_single_question()
            sq = res["cpu_baseline"]["single_question"]
            res["summary"]["cpu_full_model_1_question_ms"] = min(v for k, v in sq.items() if k.endswith("_ms"))
        else:
            res["cpu_baseline"] = None
        validate_line(res, full=world == 1 and not cfg5 and not (args.no_cpu_baseline or args.no_full_model or args.no_cfg5 or args.no_mixed
                                                    or args.no_sustained))
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
