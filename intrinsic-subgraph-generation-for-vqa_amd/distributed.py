"""Data-parallel sharding of a PyG-style batch by graph, and the logits all-gather.

The reference scales out with DDP only: every rank owns an independent Batch with LOCAL node / graph
indices (main.py:72-94, datasets/build.py:44-49).  Inference needs no gradient exchange, so the only
collective of this path is one all-gather of answer logits [B_local, 1842] per step (RCCL over xGMI when
the process group backend is "nccl"; gloo in the CPU tests).  Because of reference quirks Q1/Q3/Q4 a
shard's result is defined as the CPU path run on that shard ALONE: shards are re-indexed
locally and nothing else crosses ranks.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from .synthetic import Workload


def graph_ranges(nodes_per_graph: Tensor, edges_per_graph: Optional[Tensor], world: int, balance: bool = False
                 ) -> List[Tuple[int, int]]:
    """Contiguous graph ranges [lo, hi) per rank.  balance=False: equal graph counts (cfg4);
    balance=True: equalise sum(nodes + edges) per rank (cfg5, skewed graphs)."""
    B = nodes_per_graph.numel()
    if not balance or edges_per_graph is None:
        step = (B + world - 1) // world
        return [(min(r * step, B), min((r + 1) * step, B)) for r in range(world)]
    cost = (nodes_per_graph + edges_per_graph).double().cumsum(0)
    total = float(cost[-1]) if B else 0.0
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(torch.searchsorted(cost, torch.tensor(target, dtype=torch.double)))   # graph holding the target
        below = float(cost[i - 1]) if i > 0 else 0.0
        above = float(cost[i]) if i < B else total
        c = i if (target - below) <= (above - target) else i + 1                       # nearer boundary
        cuts.append(min(max(c, cuts[-1]), B))
    cuts.append(B)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_workload(wl: Workload, rank: int, world: int, balance: bool = False) -> Workload:
    """Rank `rank`'s contiguous graph range of `wl`, re-indexed locally (works on any device)."""
    B = wl.glf.size(0)
    npg = torch.bincount(wl.batch, minlength=B)
    eg = wl.batch[wl.edge_index[1]]
    epg = torch.bincount(eg, minlength=B)
    lo, hi = graph_ranges(npg.cpu(), epg.cpu(), world, balance)[rank]
    ptr = torch.zeros(B + 1, dtype=torch.long, device=wl.batch.device)
    ptr[1:] = npg.cumsum(0)
    n_lo, n_hi = int(ptr[lo]), int(ptr[hi])
    emask = (eg >= lo) & (eg < hi)
    return Workload(x=wl.x[n_lo:n_hi].contiguous(), edge_index=(wl.edge_index[:, emask] - n_lo).contiguous(),
                    edge_attr=wl.edge_attr[emask].contiguous(), batch=(wl.batch[n_lo:n_hi] - lo).contiguous(),
                    instr=wl.instr[:, lo:hi].contiguous(), glf=wl.glf[lo:hi].contiguous(), num_graphs=hi - lo,
                    max_nodes=int(npg[lo:hi].max()) if hi > lo else 0,
                    max_edges=int(epg[lo:hi].max()) if hi > lo else 0)


def all_gather_logits(logits: Tensor, out: Optional[Tensor] = None, group=None, async_op: bool = False):
    """[B_local, A] on every rank -> [world*B_local, A] (equal shard sizes) via all_gather_into_tensor.

    async_op=True returns (out, work): the collective runs on the communicator's own stream behind the producer of
    `logits`, so the next batch's kernels overlap it; call work.wait() before reading `out` or reusing either buffer
    (keep `logits` referenced until then)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return (logits, None) if async_op else logits
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * logits.size(0),) + tuple(logits.shape[1:]), dtype=logits.dtype,
                          device=logits.device)
    work = dist.all_gather_into_tensor(out, logits.contiguous(), group=group, async_op=async_op)
    return (out, work) if async_op else out


def all_gather_logits_ragged(logits: Tensor, group=None) -> Tensor:
    """Unequal shard sizes (balanced partitions): gather sizes, pad to the maximum, gather, trim."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return logits
    world = dist.get_world_size(group)
    n = torch.tensor([logits.size(0)], dtype=torch.long, device=logits.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    pad[: logits.size(0)] = logits
    out = torch.empty((world * m,) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)])


class GatherPipeline:
    """The per-step collective of the N > 1 path with up to `depth` all-gathers in flight (option 2): step i's
    gather runs on the communicator's stream into buffer i % (depth + 1) while the following steps' kernels run, so a
    collective up to `depth` steps long (a ring over per-link-bound xGMI) stays hidden.  `what` = "logits": BASELINE
    north_star's collective, fp32 [B_local, A] per rank (main.py:72-94 is the reference's DDP shape); "answers": the arg-max
    answers [B_local] i64 only (what the reference's evaluation reduces, utils/misc.py:40-48) -- opt-in.

    ragged=True (BASELINE configs[4]: partitions balanced by sum(nodes + edges) hold different graph counts): the ranks' row
    counts are exchanged ONCE, here; every step then gathers rows padded to the largest count (a padded staging buffer per
    in-flight slot, its pad rows zero) and `rows(buf)` trims the result.  force_collective=True issues the collective on a
    one-rank group too (tests/test_gpu_rccl_single_rank.py: this class against RCCL on one MI355X)."""

    def __init__(self, b_local: int, answers: int, device, what: str = "logits", depth: int = 2, group=None,
                 ragged: bool = False, force_collective: bool = False):
        import torch.distributed as dist
        if what not in ("logits", "answers"):
            raise ValueError(f"GatherPipeline: what = {what!r}")
        self.what, self.depth, self.group = what, max(1, int(depth)), group
        up = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if up else 1
        self.active = self.world > 1 or (bool(force_collective) and up)
        self.b_local, self.answers, self.ragged = int(b_local), int(answers), bool(ragged)
        self.sizes = [self.b_local] * self.world
        if self.ragged and self.active:
            n = torch.tensor([self.b_local], dtype=torch.long, device=device)
            got = [torch.zeros_like(n) for _ in range(self.world)]
            dist.all_gather(got, n, group=group)
            self.sizes = [int(v) for v in got]
        elif self.active and self.world > 1:          # equal shards are the caller's promise: hold it to that, once
            n = torch.tensor([self.b_local], dtype=torch.long, device=device)
            lo, hi = n.clone(), n.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if int(lo) != int(hi):
                raise ValueError(f"GatherPipeline: shards of {int(lo)}..{int(hi)} rows need ragged=True")
        self.b_max = max(self.sizes)
        shape = (self.world * self.b_max, self.answers) if what == "logits" else (self.world * self.b_max,)
        dtype = torch.float32 if what == "logits" else torch.int64
        self.buffers = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.depth + 1)] if self.active else []
        self.stage = ([torch.zeros(shape[:0] + (self.b_max,) + shape[1:], dtype=dtype, device=device) for _ in range(self.depth + 1)]
                      if self.active and self.b_local < self.b_max else [])
        self.pending = []          # (work, input kept alive), oldest first

    def drain(self, keep: int = 0) -> None:
        while len(self.pending) > keep:
            work, _keep = self.pending.pop(0)
            work.wait()

    def submit(self, i: int, logits: Tensor) -> Tensor:
        """Queue step i's gather behind the producer of `logits`; returns the buffer it lands in (valid after drain())."""
        if not self.active:
            return logits
        if logits.size(0) != self.b_local:
            raise ValueError(f"GatherPipeline.submit: {logits.size(0)} rows, built for {self.b_local}")
        self.drain(self.depth - 1)     # the oldest gather's buffer (and staging slot) is free again, its input may be released
        src = logits if self.what == "logits" else logits.argmax(dim=1)
        if self.stage:
            st = self.stage[i % (self.depth + 1)]
            st[: self.b_local].copy_(src)          # on the producer's stream; rows beyond b_local stay zero
            src = st
        import torch.distributed as dist
        out = self.buffers[i % (self.depth + 1)]
        work = dist.all_gather_into_tensor(out, src.contiguous(), group=self.group, async_op=True)
        self.pending.append((work, src))
        return out

    def rows(self, buf: Tensor) -> List[Tensor]:
        """The ranks' rows of a gathered buffer (views; rank r's padding trimmed)."""
        if not self.active:
            return [buf]
        return [buf[r * self.b_max: r * self.b_max + self.sizes[r]] for r in range(self.world)]

    def describe(self) -> dict:
        row = self.answers * 4 if self.what == "logits" else 8
        per = self.b_max * row
        d = {"collective": f"all_gather_into_tensor(logits[B_local,{self.answers}] f32)" if self.what == "logits"
             else "all_gather_into_tensor(answers[B_local] i64)",
             "bytes_per_rank": per, "bytes_received_per_rank": (self.world - 1) * per, "in_flight": self.depth}
        if self.ragged:
            d.update(ragged=True, rows_per_rank=list(self.sizes), padded_rows=self.b_max,
                     payload_bytes_per_rank=[n * row for n in self.sizes])
        return d
