"""N>1 host logic on CPU: graph sharding + logits all-gather over gloo with world_size 2.
The per-rank compute is the CPU oracle (tests may use it as the checker; the product path needs the GPU)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg(balance_case=False):
    from isubgvqa_amd import synthetic
    if balance_case:
        return synthetic.WorkloadConfig(num_graphs=24, channels=16, layers=2, masks=(1.0, 1.0), nodes_dist="pareto",
                                        nodes_min=8, nodes_max=120, edges_per_graph=0.0, degree="powerlaw", seed=8)
    return synthetic.WorkloadConfig(num_graphs=16, channels=16, layers=2, masks=(1.0, 1.0), nodes_mean=8, nodes_std=3,
                                    nodes_min=2, nodes_max=16, edges_per_graph=20, seed=5)


def _oracle_logits(cfg, wl, sd):
    from oracle import model as OM
    ocfg = OM.PathConfig(heads=cfg.heads, masking_thresholds=list(cfg.masks), use_topk=True, sampler_type=cfg.sampler,
                         sample_k=cfg.sample_k)
    with torch.no_grad():
        return OM.mgat_pool_classify(sd, wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.instr, wl.glf, ocfg)[0]


def _worker(rank, world, port, balance, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from isubgvqa_amd import synthetic
    from isubgvqa_amd.distributed import GatherPipeline, all_gather_logits, all_gather_logits_ragged, shard_workload
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _cfg(balance)
    wl = synthetic.make_workload(cfg)
    sd = {k: v.detach().clone() for k, v in synthetic.build_answer_model(cfg).state_dict().items()}
    shard = shard_workload(wl, rank, world, balance=balance)
    local = _oracle_logits(cfg, shard, sd)
    full = all_gather_logits_ragged(local) if balance else all_gather_logits(local)
    # the light collective (bench.py --gather answers): every rank's arg-max answers, async like the logits
    ans, work = (None, None) if balance else all_gather_logits(local.argmax(dim=1), async_op=True)
    if work is not None:
        work.wait()
    # bench.py's per-step collective with its DEFAULT arguments (north_star: the logits; two gathers in flight, three buffers)
    import bench
    args = bench.parse([])
    piped = None
    if not balance:
        pipe = GatherPipeline(local.size(0), local.size(1), local.device, what=args.gather, depth=args.gather_depth)
        outs = [pipe.submit(i, local + float(i)) for i in range(5)]
        pipe.drain()
        # buffers rotate over depth + 1 = 3: steps 2, 3, 4 are the ones still held
        piped = (pipe.describe(), [o.clone() for o in outs[2:]], outs[4].data_ptr() == outs[1].data_ptr())
    else:
        # BASELINE configs[4]'s collective as `bench.py --workload cfg5` builds it: balanced shards hold different graph counts, the
        # pipeline exchanges the counts once and gathers rows padded to the largest shard; an equal-shard pipeline must refuse them
        a5 = bench.parse(["--workload", "cfg5"])
        assert a5.graphs == 2048 and a5.features == "fp16" and a5.gather == "logits"
        try:
            GatherPipeline(local.size(0), local.size(1), local.device, what=a5.gather, depth=a5.gather_depth)
            refused = False
        except ValueError:
            refused = True
        pipe = GatherPipeline(local.size(0), local.size(1), local.device, what=a5.gather, depth=a5.gather_depth, ragged=True)
        outs = [pipe.submit(i, local + float(i)) for i in range(5)]
        pipe.drain()
        piped = (pipe.describe(), [torch.cat(pipe.rows(o)).clone() for o in outs[2:]], refused)
    if rank == 0:
        q.put((full, shard.num_graphs, ans, piped))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("balance", [False, True])
def test_shard_compute_allgather_world2(balance):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, balance, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, n0, ans, piped = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from isubgvqa_amd import synthetic
    cfg = _cfg(balance)
    wl = synthetic.make_workload(cfg)
    sd = {k: v.detach().clone() for k, v in synthetic.build_answer_model(cfg).state_dict().items()}
    ref = _oracle_logits(cfg, wl, sd)              # unsharded: equal because no layer samples (no Q1/Q3 coupling)
    assert full.shape == ref.shape
    assert torch.allclose(full, ref, atol=1e-5)
    if not balance:
        assert ans.dtype == torch.int64 and torch.equal(ans, full.argmax(dim=1))
        desc, held, rotated = piped
        a = full.size(1)
        assert desc["collective"] == f"all_gather_into_tensor(logits[B_local,{a}] f32)", desc       # the default IS the logits
        assert desc["in_flight"] == 2 and desc["bytes_per_rank"] == n0 * a * 4 and desc["bytes_received_per_rank"] == n0 * a * 4
        assert rotated
        for i, got in zip((2, 3, 4), held):
            assert torch.allclose(got, ref + float(i), atol=1e-5)
    else:
        desc, held, refused = piped
        assert refused, "an equal-shard GatherPipeline accepted shards of different sizes"
        sizes = desc["rows_per_rank"]
        assert desc["ragged"] and sum(sizes) == cfg.num_graphs and sizes[0] == n0 and sizes[0] != sizes[1], desc
        assert desc["padded_rows"] == max(sizes) and desc["bytes_per_rank"] == max(sizes) * full.size(1) * 4
        for i, got in zip((2, 3, 4), held):
            assert got.shape == ref.shape and torch.allclose(got, ref + float(i), atol=1e-5)


def test_graph_ranges_balance_by_nodes_plus_edges():
    from isubgvqa_amd.distributed import graph_ranges
    nodes = torch.tensor([100, 1, 1, 1, 1, 1, 1, 1])
    edges = torch.tensor([300, 2, 2, 2, 2, 2, 2, 2])
    even = graph_ranges(nodes, edges, 2, balance=False)
    assert even == [(0, 4), (4, 8)]
    bal = graph_ranges(nodes, edges, 2, balance=True)
    assert bal == [(0, 1), (1, 8)]                      # the hub graph alone outweighs the other seven
    bal4 = graph_ranges(torch.full((8,), 10), torch.full((8,), 20), 4, balance=True)
    assert bal4 == [(0, 2), (2, 4), (4, 6), (6, 8)]
    covered = [g for a, b in graph_ranges(nodes, edges, 3, balance=True) for g in range(a, b)]
    assert covered == list(range(8))                    # every graph assigned exactly once, in order


def test_shard_workload_reindexes_locally():
    from isubgvqa_amd import synthetic
    from isubgvqa_amd.distributed import shard_workload
    cfg = _cfg()
    wl = synthetic.make_workload(cfg)
    parts = [shard_workload(wl, r, 4) for r in range(4)]
    assert sum(p.num_graphs for p in parts) == cfg.num_graphs
    assert sum(p.x.size(0) for p in parts) == wl.x.size(0)
    assert sum(p.edge_index.size(1) for p in parts) == wl.edge_index.size(1)
    for p in parts:
        assert p.batch.min() == 0 and int(p.batch.max()) == p.num_graphs - 1
        assert p.edge_index.min() >= 0 and p.edge_index.max() < p.x.size(0)
        assert p.max_nodes == int(torch.bincount(p.batch).max())
