"""Tensor-level operators over the C ABI (include/isg.h): one Python function per kernel.

PyTorch is plumbing here: it owns device memory and the stream; every operator validates its
operands on the host (device, dtype, contiguity, shapes -- a wrong shape must never reach a
kernel) and then hands raw pointers to libisg_hip.so on torch's current stream.  When autograd is
recording through an operand the call is routed through autograd.py; an operator
without a backward refuses such an operand loudly.
"""
from __future__ import annotations

import contextlib
import copy
import dataclasses
import sys
import types
import weakref
from dataclasses import dataclass
from typing import NamedTuple, Optional, Tuple

import torch
from torch import Tensor

from . import _lib

MAX_NODES_PER_GRAPH = 1024   # LDS strip / sampler row capacity of the kernels


# ---------------------------------------------------------------------------------------------------------------------------------
# Every A/B switch of this module in ONE frozen object.  The functions below read `CFG.<field>`; nothing else in the module is
# mutable configuration.  `ops.SPLIT_FORWARD = False` (tests, tools, bench.py: the historical spelling) is routed by the module's
# __setattr__ to `CFG = dataclasses.replace(CFG, split_forward=False)` -- an atomic swap of the whole object, never a field
# written in place -- and `ops.SPLIT_FORWARD` reads the field; `with ops.configured(split_forward=False): ...` restores it.
@dataclasses.dataclass(frozen=True)
class Switches:
    plan_fused: bool = True
    bounds_to_host: bool = True
    mixed_dispatch: bool = True
    mixed_max_fraction: float = 0.12
    mixed_min_nodes: int = 60000
    mp_kernel: str = "graph"
    fuse_logits: bool = True
    fuse_logits_wide: bool = True          # ... also at head dimensions / edge widths beyond the tile shapes (C = 300, K = 300: round 5)
    fuse_tile_conv: bool = True
    fuse_layer_conv: bool = True
    split_stream: bool = True
    split_forward: bool = True
    fuse_gate: bool = True
    fuse_dense_tail: bool = True
    dense_tail_rows: int = 64
    fuse_readout: bool = True
    gemm_backend: str = "bf16x6"
    gemm_kernel: str = "auto"
    panel_min_n: int = 256
    embedding_sum: bool = True
    ln_planes: bool = True
    mp_planes: bool = True
    gather_add_planes: bool = True
    linear_multi: bool = True
    tile_heavy_first: bool = True
    mha_rows_planes: bool = True
    mha_rows_max_tq: int = 16

    f16x3_f16_out: bool = True
    f16x3_tile: bool = True
    gemm_f16x3: bool = True
    h3p: bool = True
    h3p_min_k: int = 256
    h3p_chain: bool = True
    h3p_min_m: int = 2048
    h3p_min_m_unsplit: int = 8192
    h3p_store_policy: int = -1
    linear_multi_h3p: bool = True
    skinny: bool = True
    rows_kernel_min_edges: int = 16384
    skinny_max_m: int = 1024
    skinny_max_work: int = 1 << 30

CFG = Switches()
_SWITCH_FIELDS = {f.name.upper(): f.name for f in dataclasses.fields(Switches)}


@contextlib.contextmanager
def configured(**fields):
    """Run a block under a modified copy of the switches (restored afterwards, exceptions included)."""
    global CFG
    keep = CFG
    CFG = dataclasses.replace(CFG, **fields)
    try:
        yield CFG
    finally:
        CFG = keep


class _OpsModule(types.ModuleType):
    def __getattr__(self, name):
        f = _SWITCH_FIELDS.get(name)
        if f is None:
            raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")
        return getattr(self.__dict__["CFG"], f)

    def __setattr__(self, name, value):
        f = _SWITCH_FIELDS.get(name)
        if f is None:
            super().__setattr__(name, value)
        else:
            self.__dict__["CFG"] = dataclasses.replace(self.__dict__["CFG"], **{f: value})

    def __delattr__(self, name):              # monkeypatch of a name that "did not exist": nothing to delete
        if name not in _SWITCH_FIELDS:
            super().__delattr__(name)


sys.modules[__name__].__class__ = _OpsModule

# Launches that leave this library's own dense kernels, and extra passes a missing hand-off costs; bench.py prints them
# per step ("no GEMM of the inference path runs on hipBLASLt" is then a number, not a sentence).
COUNTERS = {"torch_linear": 0, "torch_layer_norm": 0, "torch_attention": 0, "row_absmax": 0, "linear_h3p": 0, "h3p_segmented": 0, "linear_skinny": 0,
            "tile_nodes": 0, "oversize_nodes": 0}      # nodes the tile kernels took / nodes of graphs beyond a tile (mixed dispatch)


def reset_counters() -> None:
    for k in COUNTERS:
        COUNTERS[k] = 0


def counters() -> dict:
    return dict(COUNTERS)


def _ver(t: Tensor):
    """Validity stamp of a cached derivative of `t`: the autograd version counter, or -- for tensors made under
    torch.inference_mode(), which have none (reading `._version` raises) -- a constant: an in-place write to an inference
    tensor cannot be seen, the caches then rest on (identity, data_ptr, shape) alone.  The reference evaluates under
    inference_mode (run_token_coo.py:49), so this path has to work there."""
    return -1 if t.is_inference() else t._version


class KernelTimer:
    """Optional HIP-event bracket around every launch of one named kernel (bench.py uses it for the
    message-passing kernel).  Events are recorded on the stream the kernel is launched on."""

    def __init__(self):
        self.pairs = []
        self.meta = []

    def bracket(self, info):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.pairs.append((start, end))
        self.meta.append(info)
        return start, end

    def bracket3(self, info):
        """A bracket around TWO consecutive launches with an event between them (info["mid"]): durations_ms() gives the
        pair, split_ms() the two parts."""
        start, end = self.bracket(info)
        info["mid"] = torch.cuda.Event(enable_timing=True)
        return start, info["mid"], end

    def split_ms(self):
        return [(s.elapsed_time(m["mid"]), m["mid"].elapsed_time(e)) for (s, e), m in zip(self.pairs, self.meta) if "mid" in m]

    def drop_last(self):
        """The bracketed launch did not happen (unsupported shape, the caller takes another path)."""
        self.pairs.pop()
        self.meta.pop()

    def durations_ms(self):
        return [s.elapsed_time(e) for s, e in self.pairs]


MP_TIMER: Optional[KernelTimer] = None   # set by bench.py around its timed region
H3P_TIMER: Optional[KernelTimer] = None  # the same around every isg_linear_h3p launch (bench.py: the full model's dense share)


def _rec(*tensors) -> bool:
    """True when autograd is recording through any of the tensors: the call is then routed through autograd.py, whose
    Function.forward re-enters the same ops.* function with autograd off."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current stream's hipStream_t.  torch.cuda.current_stream() builds a Stream object per call (~20 calls and 0.1-0.2 ms
    of host time per step, which a host-bound step -- small batches, the sub-batch of run_split -- pays in full): the raw
    handle of the current device's current stream instead."""
    if _raw_stream is None or _get_device is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_get_device())


class _Ready:
    """When a cached device tensor is safe to read: the stream that built it and an event behind the building launches.  A hit
    from ANOTHER stream (run_split's side pass, a caller's own streams) makes that stream wait for the event -- once; the marker
    is dropped when the event has completed.  A hit from the building stream is ordered by the stream itself and costs one
    integer comparison."""
    __slots__ = ("stream", "event")

    def __init__(self, on_cuda: bool):
        self.stream, self.event = None, None
        if on_cuda and not torch.cuda.is_current_stream_capturing():
            self.stream = _stream()
            self.event = torch.cuda.Event()
            self.event.record()

    def wait(self) -> None:
        ev = self.event
        if ev is None or _stream() == self.stream:
            return
        if ev.query():
            self.event = None           # done for every stream from here on
        else:
            torch.cuda.current_stream().wait_event(ev)


def _chk(t: Optional[Tensor], name: str, dtype, shape=None, optional=False) -> int:
    if t is None:
        if optional:
            return 0
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise _lib.IsgError(f"{name} must live on the GPU (got {t.device}); this path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if torch.is_grad_enabled() and t.requires_grad:
        raise NotImplementedError(
            f"{name} requires grad but this operator has no backward (see autograd.py for the differentiable set); "
            "wrap the call in torch.no_grad() or detach the input")
    return t.data_ptr()


def _chk_rows(t: Tensor, name: str, dtype=torch.float32) -> int:
    """Like _chk for a 2-D fp32 (or fp16) tensor whose rows may be strided (columns contiguous, rows aligned to 4
    elements)."""
    if not t.is_cuda:
        raise _lib.IsgError(f"{name} must live on the GPU (got {t.device}); this path has no CPU fallback")
    align = 15 if dtype == torch.float32 else 7
    if t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1 or (t.stride(0) & 3) or (t.data_ptr() & align):
        raise ValueError(f"{name}: expected {dtype} [rows, cols] with contiguous columns and rows aligned to 4 elements")
    if torch.is_grad_enabled() and t.requires_grad:
        raise NotImplementedError(f"{name} requires grad but this operator has no backward (see autograd.py)")
    return t.data_ptr()


# GraphPlan.build trusts the caller's max_nodes / max_edges hints to stay free of a device->host sync; the true bounds
# are computed on the device anyway, so every hinted build queues an asynchronous copy of them into pinned memory and
# the comparison happens later, on the host, once the copy has landed: at the next build (polled, never waited for) or
# at check_plans().  An understated hint (which would size the LDS tables / sampler rows too small and truncate
# graphs silently) therefore raises -- one step late, but loudly and without stalling the stream.
_PENDING_HINTS = []      # (event, pinned int32[2] = true [max nodes, max edges], hinted max_nodes, hinted max_edges)
_PENDING_SIZES = []      # (event, pinned int64[2 or 3] = graphs beyond a tile / their nodes / their edges counted on the device, the hint's counts)


_CAPTURE_BOUNDS = []     # StepCapture: the pinned int32[2] a plan built inside the capture in progress hands its last kernel


def check_plans(block: bool = True) -> None:
    """Raise IsgError if a GraphPlan was built with hints smaller than the batch's true per-graph bounds.
    block=False only looks at copies that have already completed."""
    keep, bad = [], None
    for ev, host, hn, he in _PENDING_HINTS:
        if not block and not ev.query():
            keep.append((ev, host, hn, he))
            continue
        ev.synchronize()
        n_true, e_true = int(host[0]), int(host[1])
        if n_true > hn or (he is not None and e_true > he):
            bad = bad or (n_true, e_true, hn, he)
    _PENDING_HINTS[:] = keep
    keep_s, bad_s = [], None
    for ev, host, expect in _PENDING_SIZES:       # graph_sizes hints (GraphPlan._oversize_stats): counted on the device as well
        if not block and not ev.query():
            keep_s.append((ev, host, expect))
            continue
        ev.synchronize()
        got = [int(v) for v in host.tolist()]
        if got != expect:
            bad_s = bad_s or (got, expect)
    _PENDING_SIZES[:] = keep_s
    if bad_s is not None:
        raise _lib.IsgError(f"GraphPlan graph_sizes disagree with the batch: graphs beyond a tile / their nodes / their edges "
                            f"counted on the device {bad_s[0]}, from the hint {bad_s[1]}; results of that batch are invalid")
    if bad is not None:
        raise _lib.IsgError(f"GraphPlan hints understate the batch: max_nodes={bad[2]} / max_edges={bad[3]} given, "
                            f"but a graph has {bad[0]} nodes / {bad[1]} edges; results of that batch are invalid")


def _f32(t: Tensor) -> Tensor:
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# ------------------------------------------------------------------------------------------------
# Graph plan
# ------------------------------------------------------------------------------------------------
# CFG.plan_fused (ops.PLAN_FUSED): isg_graph_plan_build (6 launches) instead of isg_graph_ptr + isg_csr_build + isg_graph_edge_ptr (14): A/B switch
# CFG.bounds_to_host (ops.BOUNDS_TO_HOST): isg_graph_plan_build writes the batch's true bounds into pinned host memory (hint check without a copy)


# CFG.mixed_dispatch (ops.MIXED_DISPATCH): graphs beyond a tile go to the per-graph kernels, the rest of the batch stays on the tile kernels
# CFG.mixed_max_fraction (ops.MIXED_MAX_FRACTION): ... while at most this share of the batch's nodes sits in such graphs
# CFG.mixed_min_nodes (ops.MIXED_MIN_NODES): ... and the batch is large: the big graphs are a chain of ~60 launches of a workgroup or a few each, ~1.8 ms
                            # of HOST time per step whatever the batch.  Measured (profiles/r04_az_split_forward.txt), 4096
                            # graphs + 1 / 8 / 64 big ones: 1.84-1.91 / 2.00 / 2.12 ms (run_split, sub-batch on its own stream)
                            # against 2.35 / 2.49 / 2.63 ms with the per-graph kernels for everything (1.43 ms without big
                            # graphs); 1024 graphs + 1: 1.83 vs 0.89 ms.  Tests and tools lower both to force the mode.


class OversizeGraphs(NamedTuple):
    """The graphs of a batch that do not fit a graph tile, as a batch of their own (GraphPlan.oversize)."""
    gids: Tensor                 # int64 [G] graph ids
    nodes: Tensor                # int64 [Ns] node ids
    edges: Optional[Tensor]      # int64 [Es] original edge ids (None: the plan has no CSR)
    batch: Tensor                # int64 [Ns] local graph index
    edge_index: Optional[Tensor]  # int64 [2, Es] local node ids
    plan: "GraphPlan"


@dataclass
class GraphPlan:
    """What every layer needs to know about one PyG Batch, computed once on the device.

    ptr[B+1] node range per graph, nmax (device scalar + host bound), CSR by destination
    (rowptr[N+1], eid[E] original edge ids, src[E] source ids).  Replaces the reference's per-call
    ``batch[-1].item()+1``, ``to_dense_batch`` counting and PyG's per-layer index handling
    (masking.py:135,162; att_pooling.py:60; mgat_v2_conv.py:215).
    """
    N: int
    E: int
    B: int
    ptr: Tensor
    nmax_dev: Tensor
    nmax: int
    emax: int = 0
    rowptr: Optional[Tensor] = None
    eid: Optional[Tensor] = None
    src: Optional[Tensor] = None
    dst: Optional[Tensor] = None
    eptr: Optional[Tensor] = None
    batch: Optional[Tensor] = None           # kept for the backward restatements (autograd.py)
    edge_index: Optional[Tensor] = None      # kept for the lazily built CSR by source (backward only)
    _by_src: Optional[Tuple[Tensor, Tensor, Tensor]] = None
    _slots: Optional[Tensor] = None
    _tiles: Optional[dict] = None
    _edge_planes: Optional[tuple] = None
    _oversize: Optional[dict] = None
    sizes_host: Optional[Tensor] = None           # HOST int64 [2, B] nodes / in-edges per graph when the caller's collate gave them (a hint
                                                  # like max_nodes / max_edges: lets oversize() count without a device-to-host sync)
    no_tiles: bool = False                        # the plan of a batch's oversize graphs: the tile kernels are not asked again
    graph_ids: Optional[Tensor] = None            # int32 [B]: this plan's graphs are a CUT of a larger batch and these are their numbers
                                                  # there -- the samplers' in-kernel noise is keyed by them (ops._gid_ptr)
    holes: Optional["OversizeGraphs"] = None      # set by run_split: the tile kernels pass over these graphs and NOTHING fills their rows
    _memo: Optional[dict] = None                  # answers that depend on the plan and the switches only (tile_mode, a layer's dispatch):
                                                  # asked ~17 times per step by the layers, computed once (shared by run_split's copy)

    def memo(self) -> dict:
        if self._memo is None:
            self._memo = {}
        return self._memo

    def edge_planes(self, edge_attr: Tensor) -> Tuple[Tensor, Tensor]:
        """(planes int16 [E, 2, 128], inv_scale fp32 [E]) of the batch's edge features in CSR slot order (isg_edge_planes): the
        operand of isg_gatv2_tile_conv.  Every layer and head reads the same edge features (mgat.py:144-148), so the split is
        made once per batch and kept on the plan (keyed by the tensor's identity, storage and version)."""
        self.require_csr()
        key = (id(edge_attr), edge_attr.data_ptr(), _ver(edge_attr), tuple(edge_attr.shape))
        hit = self._edge_planes
        if hit is None or hit[0] != key or (len(hit) > 3 and hit[3]() is not edge_attr):    # the id of a freed tensor can be reused
            lib = _lib.load()
            E, K = edge_attr.shape
            planes = torch.empty(max(E, 1), 2, 128, dtype=torch.int16, device=edge_attr.device)
            inv = torch.empty(max(E, 1), dtype=torch.float32, device=edge_attr.device)
            _lib.check(lib.isg_edge_planes(_chk_rows(edge_attr, "edge_attr"), edge_attr.stride(0), self.eid.data_ptr(), E, K,
                                           planes.data_ptr(), inv.data_ptr(), _stream()), "isg_edge_planes")
            hit = (key, planes, inv, weakref.ref(edge_attr))
            self._edge_planes = hit
        return hit[1], hit[2]

    def tiles_and_edge_planes(self, edge_attr: Tensor, node_cap: int, edge_cap: int):
        """(tiles(node_cap, edge_cap), edge_planes(edge_attr)); when neither exists yet they are made by ONE launch
        (isg_tile_plan_edge_planes: the one-workgroup tile plan runs beside the row split instead of alone on the chip)."""
        key = (int(node_cap), int(edge_cap))
        ekey = (id(edge_attr), edge_attr.data_ptr(), _ver(edge_attr), tuple(edge_attr.shape))
        have_t = self._tiles is not None and key in self._tiles
        have_e = self._edge_planes is not None and self._edge_planes[0] == ekey
        if CFG.plan_fused and not have_t and not have_e and edge_cap > 0 and edge_attr.dim() == 2 and edge_attr.size(1) <= 128 \
                and edge_attr.size(1) % 4 == 0 and edge_attr.dtype == torch.float32:
            lib = _lib.load()
            self.require_csr()
            E, K = edge_attr.shape
            cap = int(lib.isg_tile_plan_capacity(self.N, self.E, self.B, key[0], key[1]))
            buf = torch.empty(9 * cap + 8, dtype=torch.int32, device=self.ptr.device)     # info first: 16-byte aligned
            info, heavy, tp, nt = buf[:4 * cap], buf[4 * cap:8 * cap], buf[8 * cap + 4:9 * cap + 5], buf[9 * cap + 5:9 * cap + 6]
            planes = torch.empty(max(E, 1), 2, 128, dtype=torch.int16, device=edge_attr.device)
            inv = torch.empty(max(E, 1), dtype=torch.float32, device=edge_attr.device)
            rc = lib.isg_tile_plan_edge_planes(self.ptr.data_ptr(), self.eptr.data_ptr(), self.B, key[0], key[1], tp.data_ptr(),
                                               nt.data_ptr(), info.data_ptr(), cap, heavy.data_ptr(), _chk_rows(edge_attr, "edge_attr"),
                                               edge_attr.stride(0), self.eid.data_ptr(), E, K, planes.data_ptr(), inv.data_ptr(),
                                               _stream())
            if rc != ISG_EUNSUPPORTED:
                _lib.check(rc, "isg_tile_plan_edge_planes")
                if self._tiles is None:
                    self._tiles = {}
                self._tiles[key] = (tp, nt, cap, info.view(cap, 4), heavy.view(cap, 4))
                self._edge_planes = (ekey, planes, inv, weakref.ref(edge_attr))
        return self.tiles(node_cap, edge_cap), self.edge_planes(edge_attr)

    def tiles_heavy_first(self, node_cap: int = 64, edge_cap: int = 0) -> Tensor:
        """tile_info of tiles(node_cap, edge_cap) ordered by descending CSR-slot count (32-slot classes, ties in tile order): the list
        the PERSISTENT tile kernels (isg_gatv2_layer_conv / _tile_conv: workgroup w takes entries w, w + G, ...) are handed, so that
        every workgroup gets one tile of each weight class per round -- the slowest workgroup's share of the work is 1.03x the mean
        instead of 1.06x at BASELINE configs[1] (tools/sim_tile_balance.py).  Any order gives the same results."""
        self.tiles(node_cap, edge_cap)
        hit = self._tiles[(int(node_cap), int(edge_cap))]
        return hit[4] if CFG.tile_heavy_first else hit[3]

    def tiles(self, node_cap: int = 64, edge_cap: int = 0) -> Tuple[Tensor, Tensor, int, Tensor]:
        """(tile_ptr int32[cap + 1], ntiles int32[1] on the device, cap, tile_info int32[cap, 4]): consecutive graphs packed greedily into tiles of
        at most `node_cap` nodes (and `edge_cap` CSR slots when > 0) -- the M-tiles of the fused per-layer kernels
        (csrc/isg_layer_tile.hip).  Tile t owns graphs tile_ptr[t] .. tile_ptr[t + 1] and tile_info[t] = (first node, nodes,
        first CSR slot, CSR slots); the count stays on the device (no sync): kernels are launched with `cap` workgroups, the ones
        beyond *ntiles return at once, or walk the tiles persistently.  Built on first use."""
        key = (int(node_cap), int(edge_cap))
        if self._tiles is None:
            self._tiles = {}
        hit = self._tiles.get(key)
        if hit is None:
            lib = _lib.load()
            if edge_cap > 0:
                self.require_csr()
            cap = int(lib.isg_tile_plan_capacity(self.N, self.E, self.B, key[0], key[1]))
            buf = torch.empty(9 * cap + 8, dtype=torch.int32, device=self.ptr.device)     # info first: 16-byte aligned
            info, heavy, tp, nt = buf[:4 * cap], buf[4 * cap:8 * cap], buf[8 * cap + 4:9 * cap + 5], buf[9 * cap + 5:9 * cap + 6]
            _lib.check(lib.isg_tile_plan(self.ptr.data_ptr(), self.eptr.data_ptr() if edge_cap > 0 else 0, self.B, key[0],
                                         key[1], tp.data_ptr(), nt.data_ptr(), info.data_ptr(), cap, heavy.data_ptr(), _stream()),
                       "isg_tile_plan")
            hit = (tp, nt, cap, info.view(cap, 4), heavy.view(cap, 4))
            self._tiles[key] = hit
        return hit[:4]

    def tile_mode(self, node_cap: int = 64, edge_cap: int = 256) -> str:
        """How the graph-tile kernels (isg_gatv2_layer_conv / _tile_conv, isg_mgat_dense_tail, isg_readout_tile) can take this
        batch: "tiles" -- every graph fits a tile; "mixed" -- a few do not: the tile kernels pass over them (isg_tile_plan
        gives such a graph an empty tile) and the per-graph kernels run on the list of them (`oversize`), both writing disjoint
        rows of the same outputs; "none" -- tiles do not pay (most nodes sit in oversize graphs) or the list cannot be made
        (inside a hipGraph capture: it takes a device-to-host sync)."""
        ecap = edge_cap if self.rowptr is not None else 0
        if self.B <= 0 or self.nmax <= 0 or self.no_tiles:
            return "none"
        if self.nmax <= node_cap and (ecap == 0 or self.emax <= ecap):
            return "tiles"
        if not CFG.mixed_dispatch or torch.cuda.is_current_stream_capturing():
            return "none"
        if self.N < CFG.mixed_min_nodes:
            return "none"                   # decided before the device-to-host sync below: a small batch never pays for it
        key = ("tile_mode", node_cap, ecap, CFG.mixed_max_fraction)
        hit = self.memo().get(key)
        if hit is None:
            st = self._oversize_stats(node_cap, edge_cap)
            if st is None or st["stats"][0] == 0:
                hit = "tiles"               # the hints overstated the batch
            else:   # (the LIST of such graphs -- ~30 launches -- is only built for a batch that then uses it)
                hit = "mixed" if st["stats"][1] <= CFG.mixed_max_fraction * self.N else "none"
            self._memo[key] = hit
        return hit

    def _oversize_stats(self, node_cap: int, edge_cap: int) -> Optional[dict]:
        """How many graphs of the batch lie beyond a tile, with their node / edge totals and maxima: ONE device-to-host sync, paid
        only by batches whose bounds say there is such a graph; cached per (plan, caps)."""
        ecap = int(edge_cap) if self.rowptr is not None else 0
        key = ("stats", int(node_cap), ecap)
        if self._oversize is None:
            self._oversize = {}
        if key in self._oversize:
            return self._oversize[key]
        res = None
        if self.nmax > int(node_cap) or (ecap > 0 and self.emax > ecap):
            ptr = self.ptr.long()
            n = ptr[1:] - ptr[:-1]
            big = n > int(node_cap)
            e = eptr = None
            if ecap > 0:
                eptr = self.eptr.long()
                e = eptr[1:] - eptr[:-1]
                big = big | (e > ecap)
            if self.sizes_host is not None:       # the collate's per-graph counts: no sync (verify_hints checks the bounds they imply)
                nh, eh = self.sizes_host[0], self.sizes_host[1]
                bh = nh > int(node_cap)
                if ecap > 0:
                    bh = bh | (eh > ecap)
                st = [int(bh.sum()), int(nh[bh].sum()), int(nh[bh].max()) if bool(bh.any()) else 0]
                if e is not None:
                    st += [int(eh[bh].sum()), int(eh[bh].max()) if bool(bh.any()) else 0]
                if not torch.cuda.is_current_stream_capturing():
                    # the same counts on the device, copied to pinned memory behind the stream and compared at check_plans():
                    # a wrong hint would size the lists below wrongly without any fault
                    dev_cnt = torch.stack([big, big * n] + ([big * e] if e is not None else [])).sum(1)
                    host = torch.empty(dev_cnt.numel(), dtype=torch.int64, pin_memory=True)
                    host.copy_(dev_cnt, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    _PENDING_SIZES.append((ev, host, [st[0], st[1]] + ([st[3]] if e is not None else [])))
            else:
                zero = n.new_zeros(())
                nb = torch.where(big, n, zero)
                stats = [big.sum(), nb.sum(), nb.max()]
                if e is not None:
                    eb = torch.where(big, e, zero)
                    stats += [eb.sum(), eb.max()]
                st = [int(v) for v in torch.stack(stats).tolist()]
            res = {"stats": st, "ptr": ptr, "n": n, "big": big, "eptr": eptr, "e": e}
        self._oversize[key] = res
        return res

    def oversize(self, node_cap: int = 64, edge_cap: int = 256) -> Optional["OversizeGraphs"]:
        """The graphs of this batch beyond a tile (more than node_cap nodes or edge_cap in-edges) as a batch of their own: graph
        ids, node ids, original edge ids (all ascending: segment sums keep their order), local batch vector / edge_index and the
        GraphPlan over them.  None when every graph fits.  Built once per (plan, caps); one device-to-host sync, paid only by
        batches that have such graphs.  The reference puts no cap on the objects of a scene graph (datasets/scene_graph.py:199-389)."""
        ecap = int(edge_cap) if self.rowptr is not None else 0
        key = (int(node_cap), ecap)
        if self._oversize is None:
            self._oversize = {}
        if key in self._oversize:
            return self._oversize[key]
        res = None
        st = self._oversize_stats(node_cap, edge_cap)        # ONE device-to-host sync: how many such graphs, their nodes / edges
        if st is not None:
            dev = self.ptr.device
            stats, ptr, n, big, eptr, e = st["stats"], st["ptr"], st["n"], st["big"], st["eptr"], st["e"]
            G = stats[0]
            if G > 0:
                # flags -> compaction: every list comes out ascending (segment sums keep their order) without a sort
                Ns, nmax_s = stats[1], stats[2]
                gids = torch.nonzero_static(big, size=G).squeeze(1)
                node_big = big[self.batch]                                    # [N]: the node sits in such a graph
                nodes = torch.nonzero_static(node_big, size=Ns).squeeze(1)
                seg = (big.cumsum(0) - 1)[self.batch[nodes]]                  # local graph index of those nodes
                edges = sub_ei = None
                emax_s = None
                if e is not None:
                    Es, emax_s = stats[3], stats[4]
                    edges = torch.nonzero_static(node_big[self.edge_index[1]], size=Es).squeeze(1)      # by destination: edges stay
                    newid = node_big.cumsum(0) - 1                                                    # inside their graph
                    sub_ei = newid[self.edge_index[:, edges]].contiguous()
                sub_plan = GraphPlan.build(seg.contiguous(), sub_ei, num_graphs=G, max_nodes=nmax_s, max_edges=emax_s)
                sub_plan.no_tiles = True
                own = self.graph_ids                        # a cut of a cut keeps the ORIGINAL batch's numbers
                sub_plan.graph_ids = (gids if own is None else own.long()[gids]).to(torch.int32).contiguous()
                res = OversizeGraphs(gids, nodes, edges, seg, sub_ei, sub_plan)
        self._oversize[key] = res
        return res

    def source_csr(self) -> Tuple[Tensor, Tensor, Tensor]:
        """(rowptr_s[N+1], eid_s[E], dst_s[E]): out-edges of every node in edge-id order.  Only the backward of the
        message passing needs it (d x_l is a scatter by source), so it is built on first use."""
        self.require_csr()
        if self._by_src is None:
            lib = _lib.load()
            dev = self.rowptr.device
            flipped = self.edge_index.flip(0).contiguous()
            rowptr_s = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
            eid_s = torch.empty(max(self.E, 1), dtype=torch.int32, device=dev)
            dst_s = torch.empty(max(self.E, 1), dtype=torch.int32, device=dev)
            ws_bytes = lib.isg_csr_workspace_bytes(self.N, self.E)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.isg_csr_build(flipped.data_ptr(), self.N, self.E, rowptr_s.data_ptr(), eid_s.data_ptr(),
                                         dst_s.data_ptr(), 0, ws.data_ptr(), ws_bytes, _stream()), "isg_csr_build")
            self._by_src = (rowptr_s, eid_s, dst_s)
        return self._by_src

    @staticmethod
    def build(batch: Tensor, edge_index: Optional[Tensor] = None, num_graphs: Optional[int] = None,
              max_nodes: Optional[int] = None, max_edges: Optional[int] = None,
              graph_sizes: Optional[Tensor] = None) -> "GraphPlan":
        lib = _lib.load()
        _chk(batch, "batch", torch.int64)
        N = batch.numel()
        if num_graphs is None:
            num_graphs = int(batch[-1].item()) + 1 if N > 0 else 0     # the reference's own sync (masking.py:135)
        B = int(num_graphs)
        dev = batch.device
        ptr = torch.empty(B + 1, dtype=torch.int32, device=dev)
        bounds = torch.empty(2, dtype=torch.int32, device=dev)     # [max nodes per graph, max edges per graph]
        nmax_dev = bounds[:1]
        plan = GraphPlan(N=N, E=0, B=B, ptr=ptr, nmax_dev=nmax_dev, nmax=0, batch=batch)
        if graph_sizes is not None:
            if graph_sizes.device.type != "cpu" or graph_sizes.dim() != 2 or tuple(graph_sizes.shape) != (2, B):
                raise ValueError("graph_sizes: a HOST tensor [2, num_graphs] (nodes, in-edges per graph)")
            plan.sizes_host = graph_sizes.long()
        host_bounds = None
        if edge_index is not None:
            _chk(edge_index, "edge_index", torch.int64)
            if edge_index.dim() != 2 or edge_index.size(0) != 2:
                raise ValueError(f"edge_index must be [2,E], got {tuple(edge_index.shape)}")
            E = edge_index.size(1)
            plan.E = E
            plan.edge_index = edge_index
            # one allocation for the five index arrays (rows 16-byte aligned), one for the workspace
            n1, e1 = (N + 1 + 3) // 4 * 4, (max(E, 1) + 3) // 4 * 4
            idx = torch.empty(n1 + 3 * e1 + B + 1, dtype=torch.int32, device=dev)
            plan.rowptr, plan.eid = idx[:N + 1], idx[n1:n1 + max(E, 1)]
            plan.src, plan.dst = idx[n1 + e1:n1 + e1 + max(E, 1)], idx[n1 + 2 * e1:n1 + 2 * e1 + max(E, 1)]
            plan.eptr = idx[n1 + 3 * e1:n1 + 3 * e1 + B + 1]
            ws_bytes = lib.isg_csr_workspace_bytes(N, E)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            if CFG.plan_fused:
                # hinted and eager: the plan's last kernel stores the bounds into pinned host memory itself (no copy in the stream)
                if max_nodes is not None and max_edges is not None and CFG.bounds_to_host:
                    if not torch.cuda.is_current_stream_capturing():
                        host_bounds = torch.empty(2, dtype=torch.int32, pin_memory=True)
                    elif _CAPTURE_BOUNDS:          # StepCapture: pinned memory allocated BEFORE the capture; every replay rewrites it
                        host_bounds = _CAPTURE_BOUNDS[-1]
                _lib.check(lib.isg_graph_plan_build(batch.data_ptr(), edge_index.data_ptr(), N, E, B, ptr.data_ptr(),
                                                    bounds.data_ptr(), 0 if host_bounds is None else host_bounds.data_ptr(),
                                                    plan.rowptr.data_ptr(), plan.eid.data_ptr(),
                                                    plan.src.data_ptr(), plan.dst.data_ptr(), plan.eptr.data_ptr(), ws.data_ptr(),
                                                    ws_bytes, _stream()), "isg_graph_plan_build")
            else:
                bounds.zero_()
                _lib.check(lib.isg_graph_ptr(batch.data_ptr(), N, B, ptr.data_ptr(), nmax_dev.data_ptr(), _stream()),
                           "isg_graph_ptr")
                _lib.check(lib.isg_csr_build(edge_index.data_ptr(), N, E, plan.rowptr.data_ptr(), plan.eid.data_ptr(),
                                             plan.src.data_ptr(), plan.dst.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
                           "isg_csr_build")
                _lib.check(lib.isg_graph_edge_ptr(ptr.data_ptr(), plan.rowptr.data_ptr(), B, plan.eptr.data_ptr(),
                                                  bounds[1:].data_ptr(), _stream()), "isg_graph_edge_ptr")
        else:
            bounds.zero_()
            _lib.check(lib.isg_graph_ptr(batch.data_ptr(), N, B, ptr.data_ptr(), nmax_dev.data_ptr(), _stream()),
                       "isg_graph_ptr")
        if max_nodes is None or (edge_index is not None and max_edges is None):
            got = bounds.tolist()                   # one D2H sync per batch (to_dense_batch syncs per layer)
            max_nodes = got[0] if max_nodes is None else max(int(max_nodes), got[0])
            max_edges = got[1] if max_edges is None else max(int(max_edges), got[1])
        elif torch.cuda.is_current_stream_capturing():
            # Built inside a hipGraph capture: no pinned allocation, no event to query.  The true bounds stay on the device
            # (every replay rewrites them); the owner of the graph calls plan.verify_hints() after replays -- one sync, outside
            # the captured work -- and gets the same error an eager build would raise one step late.
            plan._bounds_dev = bounds
            plan._bounds_host = host_bounds           # (StepCapture reads it between replays, without a sync)
            plan._hints = (int(max_nodes), None if edge_index is None else int(max_edges))
        else:                                       # hinted: verify later, without a sync (see check_plans)
            if _PENDING_HINTS or _PENDING_SIZES:
                check_plans(block=False)
            if host_bounds is not None:
                host = host_bounds
            else:
                host = torch.empty(2, dtype=torch.int32, pin_memory=True)
                host.copy_(bounds, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            _PENDING_HINTS.append((ev, host, int(max_nodes), None if edge_index is None else int(max_edges)))
            if len(_PENDING_HINTS) > 64:
                check_plans(block=True)
        plan.nmax = int(max_nodes)
        plan.emax = int(max_edges or 0)
        if plan.nmax > MAX_NODES_PER_GRAPH:
            raise _lib.IsgError(f"graphs with more than {MAX_NODES_PER_GRAPH} nodes are unsupported (got {plan.nmax})")
        return plan

    def verify_hints(self) -> None:
        """For a plan built inside a hipGraph capture: compare the hints with the bounds the LAST replay computed (one
        device->host sync).  Raises IsgError like check_plans(); a no-op for plans built eagerly (those are verified there)."""
        b = getattr(self, "_bounds_dev", None)
        if b is None:
            return
        n_true, e_true = (int(v) for v in b.tolist())
        hn, he = self._hints
        if n_true > hn or (he is not None and e_true > he):
            raise _lib.IsgError(f"GraphPlan hints understate the batch: max_nodes={hn} / max_edges={he} given, but a graph "
                                f"has {n_true} nodes / {e_true} edges; results of that replay are invalid")

    @staticmethod
    def edges_only(edge_index: Tensor, num_nodes: int) -> "GraphPlan":
        """CSR by destination without any per-graph structure (for callers that only have edge_index)."""
        lib = _lib.load()
        _chk(edge_index, "edge_index", torch.int64)
        dev, N, E = edge_index.device, int(num_nodes), edge_index.size(1)
        plan = GraphPlan(N=N, E=E, B=0, ptr=torch.zeros(1, dtype=torch.int32, device=dev),
                         nmax_dev=torch.zeros(1, dtype=torch.int32, device=dev), nmax=0, edge_index=edge_index)
        plan.rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        plan.eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        plan.src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        ws_bytes = lib.isg_csr_workspace_bytes(N, E)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.isg_csr_build(edge_index.data_ptr(), N, E, plan.rowptr.data_ptr(), plan.eid.data_ptr(),
                                     plan.src.data_ptr(), 0, ws.data_ptr(), ws_bytes, _stream()), "isg_csr_build")
        return plan

    def dense_slots(self) -> Tensor:
        """Position of every node in the padded [B * nmax] layout of to_dense_batch (int64 [N])."""
        if self._slots is None:
            b = self.batch
            self._slots = b * self.nmax + (torch.arange(self.N, device=b.device) - self.ptr.long()[b])
        return self._slots

    def require_csr(self) -> None:
        if self.rowptr is None:
            raise ValueError("this GraphPlan was built without edge_index")


# ------------------------------------------------------------------------------------------------
# Message passing
# ------------------------------------------------------------------------------------------------
def instr_gate(x: Tensor, instr: Tensor, batch: Tensor, plan: Optional["GraphPlan"] = None) -> Tensor:
    """gelu(x * instr[batch])   (mgat_v2_conv.py:156-157).  ``plan`` selects the HIP backward (else torch recompute)."""
    if _rec(x, instr):
        from . import autograd
        return autograd.instr_gate(x, instr, batch, plan)
    lib = _lib.load()
    N, C = x.shape
    out = torch.empty_like(x)
    _lib.check(lib.isg_instr_gate(_chk(x, "x", torch.float32), _chk(instr, "instr", torch.float32, (instr.size(0), C)),
                                  _chk(batch, "batch", torch.int64, (N,)), out.data_ptr(), N, C, _stream()),
               "isg_instr_gate")
    return out


def node_to_edge_mask(mask: Tensor, edge_index: Tensor, plan: Optional[GraphPlan] = None) -> Tensor:
    """mask[src] * mask[dst]   (sampling/node_edge_masks.py:7-10).  mask [N,1] or [N] -> [E,1] / [E].
    ``plan`` (CSR by destination) is only needed when the mask requires grad."""
    if _rec(mask):
        from . import autograd
        if plan is None:
            raise ValueError("node_to_edge_mask needs the GraphPlan to differentiate (its backward walks the CSR)")
        return autograd.node_to_edge_mask(mask, edge_index, plan)
    lib = _lib.load()
    E = edge_index.size(1)
    flat = mask.reshape(-1)
    out = torch.empty(E, dtype=torch.float32, device=mask.device)
    _lib.check(lib.isg_node_to_edge_mask(_chk(flat, "mask", torch.float32), _chk(edge_index, "edge_index", torch.int64),
                                         E, out.data_ptr(), _stream()), "isg_node_to_edge_mask")
    return out.view(E, 1) if mask.dim() == 2 else out


# CFG.mp_kernel (ops.MP_KERNEL): "graph": per-graph LDS-resident kernel; "chunk": node-chunk kernel (A/B switch for bench/tests)


def gatv2_mp(x_l: Tensor, x_r: Tensor, e_proj: Tensor, att: Tensor, plan: GraphPlan, heads: int,
             bias: Optional[Tensor] = None, node_mask: Optional[Tensor] = None, edge_mask: Optional[Tensor] = None,
             negative_slope: float = 0.2, kernel: Optional[str] = None, want_rowmax: bool = False,
             want_planes: bool = False) -> Tuple[Tensor, Tensor]:
    """MaskingGATv2Conv.message + aggregate (mgat_v2_conv.py:243-279).  Returns (out[N,H*C], alpha[E,H]).
    want_rowmax (inference, fp32 rows): the kernel also writes max |out| per (node, head) and `out` carries it as
    ``out._isg_rowmax`` [N, H] -- the row scales of the fp16 three-product GEMM that reads `out` next (x_proj).
    want_planes (inference, fp32 rows, H = 4): where the flat per-graph kernel runs (the reference's C = 300), `out` comes back
    as a SEGMENTED Planes32 -- the operand of x_proj.0 on isg_linear_h3p with no fp32 copy and no split pass; anywhere else the
    fp32 rows as always."""
    if _rec(x_l, x_r, e_proj, att, bias, node_mask, edge_mask):
        from . import autograd
        return autograd.gatv2_mp(x_l, x_r, e_proj, att, plan, heads, bias, node_mask, edge_mask, negative_slope, kernel)
    lib = _lib.load()
    plan.require_csr()
    N, HC = x_l.shape
    H = int(heads)
    C = HC // H
    E = plan.E
    if N != plan.N or H * C != HC:
        raise ValueError(f"x_l {tuple(x_l.shape)} does not match plan N={plan.N} / heads={H}")
    # x_l / x_r may be column slices of one fused [N, 2*H*C] projection: rows strided, columns contiguous
    ld_l, ld_r, ld_e = x_l.stride(0), x_r.stride(0), (e_proj.stride(0) if E > 0 else HC)
    if x_l.stride(1) != 1 or x_r.stride(1) != 1 or tuple(x_r.shape) != (N, HC) or tuple(e_proj.shape) != (E, HC):
        raise ValueError("x_l / x_r must be [N, H*C] and e_proj [E, H*C], columns contiguous")
    fdt = x_l.dtype                # feature rows: fp32, or fp16 (BASELINE configs[4]; fp32 arithmetic, per-graph kernel)
    if fdt not in (torch.float32, torch.float16) or x_r.dtype != fdt or (E > 0 and e_proj.dtype != fdt):
        raise TypeError(f"x_l / x_r / e_proj must share one dtype (fp32 or fp16), got {x_l.dtype}/{x_r.dtype}/{e_proj.dtype}")
    out = torch.empty(N, HC, dtype=fdt, device=x_l.device)
    alpha = torch.empty(E, H, dtype=torch.float32, device=x_l.device)
    use_graph = (kernel or CFG.mp_kernel) == "graph" and plan.B > 0 and plan.nmax > 0
    if fdt == torch.float16 and not use_graph:
        raise _lib.IsgError("fp16 feature rows need the per-graph kernel (a GraphPlan built with edge_index)")
    timer = MP_TIMER
    if timer is not None:
        ev0, ev1 = timer.bracket({"N": N, "E": E, "H": H, "C": C, "masked": node_mask is not None or edge_mask is not None,
                                  "feat_bytes": 2 if fdt == torch.float16 else 4})
        ev0.record()
    if want_planes and CFG.mp_planes and use_graph and fdt == torch.float32 and E > 0 and H == 4 and C % 4 == 0:
        seg = 2 * C
        st = (seg + 31) // 32
        pl = torch.empty(N * 2 * st * 64, dtype=torch.int16, device=x_l.device)
        pinv = torch.empty(2, N, dtype=torch.float32, device=x_l.device)
        rc = lib.isg_gatv2_mp_fwd_planes(
            _chk_rows(x_l, "x_l", fdt), _chk_rows(x_r, "x_r", fdt), _chk_rows(e_proj, "e_proj", fdt),
            _chk(att.reshape(-1), "att", torch.float32, (HC,)),
            _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
            plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(),
            _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
            _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True),
            pl.data_ptr(), pinv.data_ptr(), alpha.data_ptr(), N, E, H, C, float(negative_slope), plan.ptr.data_ptr(),
            plan.eptr.data_ptr(), plan.dst.data_ptr(), plan.B, plan.nmax, plan.emax, ld_l, ld_r, ld_e, _stream())
        if rc != ISG_EUNSUPPORTED:
            _lib.check(rc, "isg_gatv2_mp_fwd_planes")
            if timer is not None:
                ev1.record()
            return Planes32(pl, pinv[1], N, HC, pinv[0], seg), alpha
    rowmax = None
    if want_rowmax and use_graph and fdt == torch.float32 and E > 0:
        rowmax = torch.empty(N, H, dtype=torch.float32, device=x_l.device)
        rc = lib.isg_gatv2_mp_fwd_rowmax(
            _chk_rows(x_l, "x_l", fdt), _chk_rows(x_r, "x_r", fdt), _chk_rows(e_proj, "e_proj", fdt),
            _chk(att.reshape(-1), "att", torch.float32, (HC,)),
            _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
            plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(),
            _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
            _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True),
            out.data_ptr(), alpha.data_ptr(), rowmax.data_ptr(), N, E, H, C, float(negative_slope), plan.ptr.data_ptr(),
            plan.eptr.data_ptr(), plan.dst.data_ptr(), plan.B, plan.nmax, plan.emax, ld_l, ld_r, ld_e, _stream())
        if rc == ISG_EUNSUPPORTED:
            rowmax = None              # this batch / width takes another kernel: plain call below
        else:
            _lib.check(rc, "isg_gatv2_mp_fwd_rowmax")
            if timer is not None:
                ev1.record()
            attach_row_maxima(out, rowmax)
            return out, alpha
    entry = lib.isg_gatv2_mp_fwd if fdt == torch.float32 else lib.isg_gatv2_mp_fwd_f16
    _lib.check(entry(
        _chk_rows(x_l, "x_l", fdt), _chk_rows(x_r, "x_r", fdt),
        _chk_rows(e_proj, "e_proj", fdt) if E > 0 else 0, _chk(att.reshape(-1), "att", torch.float32, (HC,)),
        _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
        plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True),
        out.data_ptr(), alpha.data_ptr(), N, E, H, C, float(negative_slope),
        plan.ptr.data_ptr() if use_graph else 0, plan.eptr.data_ptr() if use_graph else 0,
        plan.dst.data_ptr() if use_graph else 0, plan.B,
        plan.nmax if use_graph else 0, plan.emax if use_graph else 0, ld_l, ld_r, ld_e, _stream()), "isg_gatv2_mp_fwd")
    if timer is not None:
        ev1.record()
    return out, alpha


# lin_edge folded into the attention logits (csrc/isg_mp_logits.hip): e_proj [E, H*C] is never written or read
# CFG.fuse_logits (ops.FUSE_LOGITS): 
GK_NCAP_L, GK_ECAP_L = 256, 1024      # the per-graph message-passing kernel's largest tables (csrc/isg_mp_graph.hip)


def fused_logits_supported(plan: "GraphPlan", heads: int, channels: int, edge_dim: int) -> bool:
    """Shape test of isg_gatv2_edge_logits + isg_gatv2_mp_fwd_logits (inference, fp32 rows, per-graph kernel)."""
    cp = (channels + 31) // 32 * 32       # round 5: heads padded to whole 32-channel tiles (the reference's C = 300 -> 320), K <= 304
    wide = channels % 32 != 0 or edge_dim > 128
    if wide and plan.E < CFG.rows_kernel_min_edges:
        # the rows kernel streams all of lin_edge's tiles through its three-slot ring whatever the number of slots: ~94 us for 400
        # edges as for 50 000 (40 tiles x one DMA round trip each).  A small batch projects its few edge rows (isg_linear_skinny, ~5 us)
        # and runs the flat kernel on e_proj instead
        return False
    return (CFG.fuse_logits and (CFG.fuse_logits_wide or not wide) and CFG.gemm_backend == "bf16x6" and CFG.gemm_f16x3 and
            CFG.mp_kernel == "graph" and channels % 4 == 0 and heads * cp <= 2048 and 0 < edge_dim <= 304 and edge_dim % 4 == 0 and
            plan.B > 0 and plan.nmax > 0 and plan.rowptr is not None and plan.E > 0)


def _edge_logits_weight(w_edge: Tensor, heads: int):
    """lin_edge.weight [H*C, K] as the fragment planes isg_gatv2_edge_logits reads: the weight itself when 32 | C and K <= 128; else
    with zero rows behind every head's C-th up to the next multiple of 32 (C = 300 -> 320: a channel tile never straddles heads)
    and, for K > 128 (the rows kernel: 19 k steps), zero columns up to 304."""
    HC, K = w_edge.shape
    C = HC // heads
    cp = (C + 31) // 32 * 32
    kp = K if K <= 128 else 304
    if cp == C and kp == K:
        return _weight_planes(w_edge, True, "f16x3")

    def build():
        w = torch.zeros(heads, cp, kp, dtype=torch.float32, device=w_edge.device)
        w[:, :C, :K] = w_edge.detach().view(heads, C, K)
        return w.view(heads * cp, kp)
    return _weight_planes(derived_weight(f"edge_logits_pad{heads}", (w_edge,), build), True, "f16x3")


def gatv2_edge_logits(x_l: Tensor, x_r: Tensor, edge_attr: Tensor, w_edge: Tensor, att: Tensor, plan: "GraphPlan", heads: int,
                      node_mask: Optional[Tensor] = None, edge_mask: Optional[Tensor] = None,
                      negative_slope: float = 0.2) -> Optional[Tensor]:
    """isg_gatv2_edge_logits alone: logits fp32 [E, H] in CSR SLOT order (slot t = edge plan.eid[t]); None if unsupported."""
    lib = _lib.load()
    plan.require_csr()
    N, HC = x_l.shape
    H = int(heads)
    E, K = edge_attr.shape
    planes, inv = _edge_logits_weight(w_edge, H)
    logits = torch.empty(E, H, dtype=torch.float32, device=x_l.device)
    fdt = x_l.dtype                      # fp32 rows, or half rows (isg_gatv2_edge_logits_f16: BASELINE configs[4]'s storage, K >= 128)
    entry = lib.isg_gatv2_edge_logits if fdt == torch.float32 else lib.isg_gatv2_edge_logits_f16
    rc = entry(
        _chk_rows(edge_attr, "edge_attr"), edge_attr.stride(0), planes.data_ptr(), inv.data_ptr(),
        _chk_rows(x_l, "x_l", fdt), x_l.stride(0), 0, _chk_rows(x_r, "x_r", fdt), x_r.stride(0), 0,
        _chk(att.reshape(-1), "att", torch.float32, (HC,)), plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(),
        _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        logits.data_ptr(), E, H, HC // H, K, float(negative_slope), _stream())
    if rc == ISG_EUNSUPPORTED:
        return None
    _lib.check(rc, "isg_gatv2_edge_logits")
    return logits


def gatv2_mp_edge_logits(x_l: Tensor, x_r: Tensor, edge_attr: Tensor, w_edge: Tensor, att: Tensor, plan: "GraphPlan",
                         heads: int, bias: Optional[Tensor] = None, node_mask: Optional[Tensor] = None,
                         edge_mask: Optional[Tensor] = None, negative_slope: float = 0.2, want_rowmax: bool = False,
                         want_planes: bool = False):
    """gatv2_mp(x_l, x_r, lin_edge(edge_attr), ...) as two launches that never materialise lin_edge's output
    (mgat_v2_conv.py:243-279 with :259-261 inside): isg_gatv2_edge_logits forms the logits [E, H] in the epilogue of the
    edge GEMM, isg_gatv2_mp_fwd_logits does softmax + aggregation.  Returns (out, alpha), or None when the per-graph kernel
    has no instantiation for this batch / width (the caller then runs the un-fused pair)."""
    lib = _lib.load()
    plan.require_csr()
    N, HC = x_l.shape
    H = int(heads)
    C = HC // H
    E, K = edge_attr.shape
    if N != plan.N or E != plan.E or tuple(w_edge.shape) != (HC, K):
        raise ValueError("gatv2_mp_edge_logits: operand shapes do not match the plan")
    if x_l.dtype not in (torch.float32, torch.float16) or edge_attr.dtype != torch.float32:
        raise TypeError("gatv2_mp_edge_logits: x_l / x_r as fp32 or fp16 rows, fp32 edge rows")
    if tuple(x_r.shape) != (N, HC) or x_r.dtype != x_l.dtype:
        raise ValueError("gatv2_mp_edge_logits: x_r must be [N, H*C] of x_l's type")
    planes, inv = _edge_logits_weight(w_edge, H)
    if x_l.dtype == torch.float16:
        # BASELINE configs[4]'s storage (fp16 feature rows, fp32 arithmetic): the rows kernel gathers half rows and rounds the edge
        # projection to half as the un-fused path stores it; the result row leaves as half.  K >= 128 (the rows kernel only).
        if K < 128:
            return None
        logits = torch.empty(E, H, dtype=torch.float32, device=x_l.device)
        alpha = torch.empty(E, H, dtype=torch.float32, device=x_l.device)
        out = torch.empty(N, HC, dtype=torch.float16, device=x_l.device)
        nm = _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True)
        em = _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True)
        attp = _chk(att.reshape(-1), "att", torch.float32, (HC,))
        timer = MP_TIMER
        if timer is not None:
            ev0, evm, ev1 = timer.bracket3({"N": N, "E": E, "H": H, "C": C, "K": K,
                                            "masked": node_mask is not None or edge_mask is not None, "feat_bytes": 2,
                                            "fused_logits": True})
            ev0.record()
        rc = lib.isg_gatv2_edge_logits_f16(
            _chk_rows(edge_attr, "edge_attr"), edge_attr.stride(0), planes.data_ptr(), inv.data_ptr(),
            _chk_rows(x_l, "x_l", torch.float16), x_l.stride(0), 0, _chk_rows(x_r, "x_r", torch.float16), x_r.stride(0), 0, attp,
            plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(), em, nm, logits.data_ptr(), E, H, C, K,
            float(negative_slope), _stream())
        if rc == ISG_EUNSUPPORTED:
            if timer is not None:
                timer.drop_last()
            return None
        _lib.check(rc, "isg_gatv2_edge_logits_f16")
        if timer is not None:
            evm.record()
        rc = lib.isg_gatv2_mp_fwd_logits_f16(
            _chk_rows(x_l, "x_l", torch.float16), logits.data_ptr(), attp,
            _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
            plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), nm, em, out.data_ptr(), alpha.data_ptr(),
            N, E, H, C, float(negative_slope), plan.ptr.data_ptr(), plan.eptr.data_ptr(), plan.dst.data_ptr(),
            plan.B, plan.nmax, plan.emax, x_l.stride(0), _stream())
        if rc == ISG_EUNSUPPORTED:
            if timer is not None:
                timer.drop_last()
            return None
        _lib.check(rc, "isg_gatv2_mp_fwd_logits_f16")
        if timer is not None:
            ev1.record()
        return out, alpha
    logits = torch.empty(E, H, dtype=torch.float32, device=x_l.device)
    alpha = torch.empty(E, H, dtype=torch.float32, device=x_l.device)
    # the result as the segmented planes32 operand of x_proj.0 (the flat kernel, H = 4: the reference's C = 300) -- or fp32 rows
    as_planes = bool(want_planes and CFG.mp_planes and H == 4 and C % 32 != 0)
    out = None if as_planes else torch.empty(N, HC, dtype=torch.float32, device=x_l.device)
    rowmax = torch.empty(N, H, dtype=torch.float32, device=x_l.device) if want_rowmax and not as_planes and C % 32 == 0 else None
    nm = _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True)
    em = _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True)
    attp = _chk(att.reshape(-1), "att", torch.float32, (HC,))
    timer = MP_TIMER
    if timer is not None:   # bench.py: ONE bracket around both launches -- together they are the reference's message +
        # aggregate (plus lin_edge); the roofline keeps the un-fused algorithmic bytes of SURVEY 8(d)
        ev0, evm, ev1 = timer.bracket3({"N": N, "E": E, "H": H, "C": C, "K": K,
                                        "masked": node_mask is not None or edge_mask is not None, "feat_bytes": 4,
                                        "fused_logits": True})
        ev0.record()
    rc = lib.isg_gatv2_edge_logits(
        _chk_rows(edge_attr, "edge_attr"), edge_attr.stride(0), planes.data_ptr(), inv.data_ptr(),
        _chk_rows(x_l, "x_l"), x_l.stride(0), 0, _chk_rows(x_r, "x_r"), x_r.stride(0), 0, attp,
        plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(), em, nm, logits.data_ptr(), E, H, C, K,
        float(negative_slope), _stream())
    if rc == ISG_EUNSUPPORTED:
        if timer is not None:
            timer.drop_last()
        return None
    _lib.check(rc, "isg_gatv2_edge_logits")
    if timer is not None:
        evm.record()
    if as_planes:
        seg = 2 * C
        st = (seg + 31) // 32
        pl = torch.empty(N * 2 * st * 64, dtype=torch.int16, device=x_l.device)
        pinv = torch.empty(2, N, dtype=torch.float32, device=x_l.device)
        rc = lib.isg_gatv2_mp_fwd_logits_planes(
            _chk_rows(x_l, "x_l"), logits.data_ptr(), attp,
            _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
            plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), nm, em, pl.data_ptr(), pinv.data_ptr(),
            alpha.data_ptr(), N, E, H, C, float(negative_slope), plan.ptr.data_ptr(), plan.eptr.data_ptr(), plan.dst.data_ptr(),
            plan.B, plan.nmax, plan.emax, x_l.stride(0), _stream())
        if rc != ISG_EUNSUPPORTED:
            _lib.check(rc, "isg_gatv2_mp_fwd_logits_planes")
            if timer is not None:
                ev1.record()
            return Planes32(pl, pinv[1], N, HC, pinv[0], seg), alpha
        out = torch.empty(N, HC, dtype=torch.float32, device=x_l.device)
    rc = lib.isg_gatv2_mp_fwd_logits(
        _chk_rows(x_l, "x_l"), logits.data_ptr(), attp,
        _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
        plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), nm, em, out.data_ptr(), alpha.data_ptr(),
        0 if rowmax is None else rowmax.data_ptr(), N, E, H, C, float(negative_slope), plan.ptr.data_ptr(),
        plan.eptr.data_ptr(), plan.dst.data_ptr(), plan.B, plan.nmax, plan.emax, x_l.stride(0), _stream())
    if rc == ISG_EUNSUPPORTED:
        if timer is not None:
            timer.drop_last()
        return None
    _lib.check(rc, "isg_gatv2_mp_fwd_logits")
    if timer is not None:
        ev1.record()
    if rowmax is not None:
        attach_row_maxima(out, rowmax)
    return out, alpha


class StepCapture:
    """Product-path hipGraph execution (opt-in: `AnswerModel.forward(..., capture=True)`, `ISubGVQA.forward(..., capture=True)`).
    A forward over fixed-shape inputs is ~25-110 launches; below ~1000 graphs per step the step is launch-bound (1 024
    graphs 0.641 ms eager vs 0.542 ms replayed, 256 graphs 0.652 vs 0.332).  `run(fn, tensors, key)` keeps one captured hipGraph
    per KEY = (shape / dtype / device of every input, the caller's hints and options, the module's switches): the first call of a key
    runs `fn` eagerly on private static copies of the inputs (kernel attributes, derived weights, allocator pools; the plan's hints
    are checked), captures it once, and every call replays it after copying the caller's tensors into the static ones (a tensor
    that already IS the static one is not copied).  `fn(*tensors) -> (outputs, plan)`: the GraphPlan must be built INSIDE fn from
    host-side bounds (no device-to-host read is possible in a capture); its true bounds are written to pinned host memory by the
    plan's own last kernel on every replay and compared with the hints at the next call -- a batch whose graphs exceed the hints
    raises IsgError one call late, exactly like an eager hinted build (check_plans).  Outputs are the graph's static tensors: valid
    until the next call with the same key.  Nothing here is a fallback: a forward that cannot be captured raises."""

    def __init__(self, max_entries: int = 8):
        import collections
        self.entries = collections.OrderedDict()
        self.max_entries = int(max_entries)
        self.replays = 0
        self.captures = 0

    @staticmethod
    def _sig(t):
        return None if t is None else (tuple(t.shape), t.dtype, t.device.index)

    def _check_bounds(self, ent) -> None:
        plan, host = ent["plan"], ent["host"]
        if plan is None or host is None or not ent["event"].query():
            return                                      # the last replay has not finished: look again at the next call
        n_true, e_true = int(host[0]), int(host[1])
        hn, he = plan._hints
        if n_true > hn or (he is not None and e_true > he):
            raise _lib.IsgError(f"GraphPlan hints understate the batch: max_nodes={hn} / max_edges={he} given, but a graph of a "
                                f"recent replay has {n_true} nodes / {e_true} edges; results of that replay are invalid")

    def run(self, fn, tensors, key_extra=(), warm: int = 2):
        if torch.is_grad_enabled():
            raise RuntimeError("StepCapture: inference only (wrap the call in torch.no_grad() / inference_mode())")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("StepCapture: already inside a capture")
        key = (tuple(self._sig(t) for t in tensors), key_extra, CFG)
        ent = self.entries.get(key)
        if ent is None:
            static = [None if t is None else t.clone() for t in tensors]
            for _ in range(max(1, warm)):
                fn(*static)
            check_plans()                              # the eager runs' hints: a wrong one raises HERE, before anything is captured
            torch.cuda.synchronize()
            host = torch.zeros(2, dtype=torch.int32, pin_memory=True)
            graph = torch.cuda.CUDAGraph()
            _CAPTURE_BOUNDS.append(host)
            try:
                with torch.cuda.graph(graph):
                    outs, plan = fn(*static)
            finally:
                _CAPTURE_BOUNDS.pop()
            if plan is not None and getattr(plan, "_bounds_host", None) is None:
                host = None                            # (a plan without an edge list / without both hints keeps its bounds on the device)
            ent = {"graph": graph, "static": static, "outs": outs, "plan": plan, "host": host, "event": torch.cuda.Event()}
            self.entries[key] = ent
            self.captures += 1
            while len(self.entries) > self.max_entries:
                self.entries.popitem(last=False)
        else:
            self.entries.move_to_end(key)
            self._check_bounds(ent)
            for st, t in zip(ent["static"], tensors):
                if t is not None and st.data_ptr() != t.data_ptr():
                    st.copy_(t, non_blocking=True)
        ent["graph"].replay()
        ent["event"].record()
        self.replays += 1
        return ent["outs"]

    def verify(self) -> None:
        """Synchronise and check every entry's last replay (tests; the end of an evaluation loop)."""
        torch.cuda.synchronize()
        for ent in self.entries.values():
            self._check_bounds(ent)
            if ent["plan"] is not None and ent["host"] is None:
                ent["plan"].verify_hints()


ISG_EUNSUPPORTED = -2      # include/isg.h

# message + softmax + aggregation with lin_edge inside as ONE launch on graph-aligned tiles (csrc/isg_layer_tile.hip): the
# head's x_l slice of a tile is staged once in LDS and serves the logit epilogue's row gathers and the aggregation
# CFG.fuse_tile_conv (ops.FUSE_TILE_CONV): 
TILE_CONV_NODES, TILE_CONV_EDGES = 64, 256


def tile_conv_supported(plan: "GraphPlan", heads: int, channels: int, edge_dim: int) -> bool:
    """Shape test of isg_gatv2_tile_conv (inference, fp32 rows): C = 128, edge features <= 128 wide, every graph within one
    64-node / 256-slot tile -- or all but a few (GraphPlan.tile_mode: those go to the per-graph kernels)."""
    return (CFG.fuse_tile_conv and CFG.fuse_logits and CFG.gemm_backend == "bf16x6" and CFG.gemm_f16x3 and CFG.mp_kernel == "graph" and
            channels == 128 and 0 < edge_dim <= 128 and edge_dim % 4 == 0 and heads <= 64 and plan.B > 0 and
            plan.rowptr is not None and plan.E > 0 and plan.tile_mode(TILE_CONV_NODES, TILE_CONV_EDGES) != "none")


# CFG.fuse_layer_conv (ops.FUSE_LAYER_CONV): ... and lin_l | lin_r inside as well (csrc/isg_layer_conv.hip): x_l / x_r never exist in memory


def layer_conv_supported(plan: "GraphPlan", heads: int, channels: int, in_channels: int, edge_dim: int) -> bool:
    """Shape test of isg_gatv2_layer_conv: isg_gatv2_tile_conv's, and a 128-wide layer input."""
    return CFG.fuse_layer_conv and in_channels == 128 and heads <= 16 and tile_conv_supported(plan, heads, channels, edge_dim)


class NodePlanes(NamedTuple):
    """Node rows as isg_gatv2_layer_conv reads them: per-row power-of-two scale, (hi, mid) fp16 planes int16 [N, 2, 128] and the
    inverse scales fp32 [N] (the staging of the exact-split Linears, made once per row by the kernel that produces the rows)."""
    planes: Tensor
    inv: Tensor


def node_planes(x: Tensor) -> NodePlanes:
    """fp32 rows [N, 128] -> NodePlanes (isg_edge_planes in row order): for callers that hold the gated rows only as fp32."""
    lib = _lib.load()
    N, K = x.shape
    planes = torch.empty(max(N, 1), 2, 128, dtype=torch.int16, device=x.device)
    inv = torch.empty(max(N, 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.isg_edge_planes(_chk_rows(x, "x"), x.stride(0), 0, N, K, planes.data_ptr(), inv.data_ptr(), _stream()),
               "isg_edge_planes")
    return NodePlanes(planes, inv)


def instr_gate_planes(x: Tensor, instr: Tensor, batch: Tensor, want_rows: bool = False) -> Tuple[Optional[Tensor], NodePlanes]:
    """gelu(x * instr[batch]) (mgat_v2_conv.py:156-157) as NodePlanes for gatv2_layer_conv, plus the fp32 rows when a masked
    layer's node gate needs them: isg_instr_gate_planes.  Inference only (C = 128)."""
    lib = _lib.load()
    N, C = x.shape
    rows = torch.empty_like(x) if want_rows else None
    planes = torch.empty(max(N, 1), 2, 128, dtype=torch.int16, device=x.device)
    inv = torch.empty(max(N, 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.isg_instr_gate_planes(_chk(x, "x", torch.float32), _chk(instr, "instr", torch.float32, (instr.size(0), C)),
                                         _chk(batch, "batch", torch.int64, (N,)), 0 if rows is None else rows.data_ptr(),
                                         planes.data_ptr(), inv.data_ptr(), N, C, _stream()), "isg_instr_gate_planes")
    return rows, NodePlanes(planes, inv)


def _count_tile_nodes(plan: "GraphPlan", sub: Optional["OversizeGraphs"]) -> None:
    COUNTERS["tile_nodes"] += plan.N - (0 if sub is None else sub.nodes.numel())
    COUNTERS["oversize_nodes"] += 0 if sub is None else sub.nodes.numel()


def _mixed_sub(plan: "GraphPlan") -> Optional["OversizeGraphs"]:
    """The oversize graphs of a batch the tile kernels take in "mixed" mode (None in "tiles" mode) -- and None when they run as a
    batch of their own through the whole model (run_split: plan.holes), so that nobody fills their rows layer by layer."""
    if plan.holes is not None:
        _count_tile_nodes(plan, plan.holes)
        return None
    sub = plan.oversize(TILE_CONV_NODES, TILE_CONV_EDGES) if plan.tile_mode(TILE_CONV_NODES, TILE_CONV_EDGES) == "mixed" else None
    _count_tile_nodes(plan, sub)
    return sub


# CFG.split_stream (ops.SPLIT_STREAM): ... on a stream of its own, beside the whole batch's tile kernels (A/B switch)
_side_streams: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


# CFG.split_forward (ops.SPLIT_FORWARD): "mixed" batches: the graphs beyond a tile run as a batch of their own through the WHOLE model (A/B switch;
                          # off: every tile kernel's wrapper fills their rows with the per-graph kernels, layer by layer)


def oversize_split(plan: "GraphPlan") -> Optional["OversizeGraphs"]:
    """The graphs a model running on the graph-tile kernels should send through run_split (None: none, or not worth it)."""
    if not CFG.split_forward or plan.holes is not None or plan.rowptr is None:
        return None
    if plan.tile_mode(TILE_CONV_NODES, TILE_CONV_EDGES) != "mixed":
        return None
    return plan.oversize(TILE_CONV_NODES, TILE_CONV_EDGES)


def run_split(plan: "GraphPlan", sub: "OversizeGraphs", core, x: Tensor, edge_index: Tensor, edge_attr: Tensor, batch: Tensor,
              instr: Tensor, glf: Tensor, noises=None, seed=None, kinds: str = "gnn"):
    """core(x, edge_index, edge_attr, batch, instr, glf, plan, noises, seed, gate_feats) -> tuple of tensors / lists / None, one per
    letter of `kinds` ("g": a row per graph, "n": a row per node), run TWICE: on the whole batch with the tile kernels passing
    over the graphs beyond a tile (their rows stay unwritten: plan.holes), and on those graphs as a batch of their own (per-graph
    kernels; instr is [L, B, C]); the second result is then copied over the rows of the first.  Every op of the path is local to
    a graph (or a row), so nothing of a hole reaches another graph -- with ONE exception that the reference itself makes: a masked
    layer's node gate reads the question row batch[batch[n]] (masking.py:151-155, quirk Q3), i.e. for graph g the row of the graph
    that holds NODE number g; the sub-batch is handed exactly those rows (gate_feats, to MGAT.forward).  A real GQA batch has a few such graphs (the reference caps
    nothing: datasets/scene_graph.py:199-389); filling their rows after every tile kernel instead cost ~200 small launches per step
    (profiles/r04_az_split_forward.txt).  Dense [B, nmax] noise is cut to the sub-batch; under a seed every graph of the sub-batch draws the stream of its number in the WHOLE batch (plan.graph_ids), so a seeded mask does not depend on the dispatch."""
    nmax_s = sub.plan.nmax

    def run_side():
        xs, es = x.index_select(0, sub.nodes), edge_attr.index_select(0, sub.edges)
        instr_s, glf_s = instr.index_select(1, sub.gids), glf.index_select(0, sub.gids)
        gate_s = glf.index_select(0, batch.index_select(0, sub.gids.clamp(max=max(plan.N - 1, 0))))
        nz_s = None
        if noises is not None:
            nz_s = {}
            for k, v in noises.items():
                v = v.index_select(0, sub.gids)
                nz_s[k] = (v[:, :nmax_s] if v.dim() == 2 else v[:, :, :nmax_s]).contiguous()
        return core(xs, sub.edge_index, es, sub.batch, instr_s, glf_s, sub.plan, nz_s, seed, gate_s)

    def run_main():
        # the SAME tensors and caches, its own `holes`: the caller's plan is not written to, and a second run_split on it (another
        # thread, a re-entrant core) sees none.  The lazily created cache containers are made on `plan` BEFORE the copy, so that
        # what the main pass builds (tiles, edge planes, memo) lands where the caller's plan finds it again
        if plan._tiles is None:
            plan._tiles = {}
        if plan._oversize is None:
            plan._oversize = {}
        plan.memo()
        holed = copy.copy(plan)
        holed.holes = sub
        holed._memo = {}               # (a layer's dispatch depends on `holes`: the holed plan answers for itself)
        res = core(x, edge_index, edge_attr, batch, instr, glf, holed, noises, seed, None)
        plan._edge_planes = holed._edge_planes          # (a tuple, not a container: handed back)
        return res

    if CFG.split_stream and x.is_cuda and not torch.cuda.is_current_stream_capturing():
        # The sub-batch is a chain of ~60 launches of one or a few workgroups each (0.65 ms of GPU time for ONE 100-node graph):
        # on a stream of its own it runs beside the tile kernels instead of behind them.  The main pass is issued FIRST (the GPU
        # starts on it while the host is still issuing the sub-batch).
        cur = torch.cuda.current_stream()
        side_stream = _side_stream(x.device)
        side_stream.wait_stream(cur)                 # inputs and the lists of `sub` are complete
        main = run_main()                            # (weight caches the two passes share: every entry carries its _Ready marker)
        with torch.cuda.stream(side_stream):
            side = run_side()
        cur.wait_stream(side_stream)

        def keep(t):                                 # results made on the side stream, read on this one
            if isinstance(t, Tensor):
                t.record_stream(cur)
            elif isinstance(t, (list, tuple)):
                for u in t:
                    keep(u)
        keep(side)
    else:
        main = run_main()
        side = run_side()
    if len(main) != len(kinds) or len(side) != len(kinds):
        raise ValueError("run_split: one letter of `kinds` per result")

    def merge(m, s_, kind):
        if m is None:
            return None
        if isinstance(m, (list, tuple)):
            return type(m)(merge(a, b, kind) for a, b in zip(m, s_))
        idx = sub.gids if kind == "g" else sub.nodes
        if m.size(0) != (plan.B if kind == "g" else plan.N) or s_.size(0) != idx.numel():
            raise ValueError(f"run_split: a '{kind}' result of {tuple(m.shape)} / {tuple(s_.shape)} rows")
        if m.dim() == 2 and kind == "n" and s_.dim() == 2 and m.size(1) != s_.size(1):
            raise ValueError("run_split: per-node results of different widths")
        return m.index_copy_(0, idx, s_.to(m.dtype))
    return tuple(merge(m, s_, k) for m, s_, k in zip(main, side, kinds))


def _oversize_conv(sub: "OversizeGraphs", x_l: Tensor, x_r: Tensor, edge_attr: Tensor, w_edge: Tensor, att: Tensor, heads: int,
                   bias, node_mask, edge_mask, negative_slope: float, out: Tensor, alpha: Tensor, rowmax: Optional[Tensor]) -> None:
    """Message passing of the graphs the tile kernels passed over (mgat_v2_conv.py:215-279 on the sub-batch): lin_edge +
    the per-graph kernel (256-node / 1024-edge tables, or node chunks beyond), written into the rows / edges of `out` /
    `alpha` / `rowmax` that belong to those graphs."""
    key = (id(edge_attr), edge_attr.data_ptr(), _ver(edge_attr))          # every layer reads the same edge features
    hit = getattr(sub.plan, "_parent_edge_rows", None)
    if hit is None or hit[0] != key:
        hit = (key, edge_attr.index_select(0, sub.edges))
        sub.plan._parent_edge_rows = hit
    e_proj = linear(hit[1], w_edge)
    nm = None if node_mask is None else node_mask.reshape(-1).index_select(0, sub.nodes)
    em = None if edge_mask is None else edge_mask.reshape(-1).index_select(0, sub.edges)
    o, a = gatv2_mp(x_l, x_r, e_proj, att, sub.plan, heads, bias=bias, node_mask=nm, edge_mask=em,
                    negative_slope=negative_slope, want_rowmax=rowmax is not None)
    out.index_copy_(0, sub.nodes, o)
    alpha.index_copy_(0, sub.edges, a)
    if rowmax is not None:
        rm = row_maxima(o)
        if rm is None:                                  # the node-chunk kernel leaves none: one pass over the few rows
            rm = o.view(o.size(0), heads, -1).abs().amax(dim=2)
        rowmax.index_copy_(0, sub.nodes, rm)


def gatv2_layer_conv(x, lin_l, lin_r, edge_attr: Tensor, w_edge: Tensor, att: Tensor, plan: "GraphPlan", heads: int,
                     bias: Optional[Tensor] = None, node_mask: Optional[Tensor] = None, edge_mask: Optional[Tensor] = None,
                     negative_slope: float = 0.2, want_rowmax: bool = False):
    """lin_l(x), lin_r(x), lin_edge(edge_attr), message, softmax and aggregation of one MaskingGATv2Conv as ONE launch
    (mgat_v2_conv.py:177-181, :215-232, :243-279): isg_gatv2_layer_conv.  x = the gated layer input [N, 128], as NodePlanes (from
    instr_gate_planes / mgat_dense_tail) or as fp32 rows (split here, one more launch).  Bit-identical to linear_fused +
    gatv2_tile_conv.  Returns (out, alpha), or None when the kernel has no launch for this shape."""
    lib = _lib.load()
    plan.require_csr()
    if not isinstance(x, NodePlanes):
        if x.dtype != torch.float32 or x.dim() != 2 or x.size(1) != 128:
            raise TypeError("gatv2_layer_conv: fp32 rows [N, 128] or NodePlanes")
        x = node_planes(x)
    N, K_in = plan.N, 128
    if x.planes.size(0) < N or x.inv.size(0) < N:
        raise ValueError("gatv2_layer_conv: node planes shorter than the plan's node count")
    dev = x.planes.device
    H = int(heads)
    HC = lin_l.weight.size(0)
    C = HC // H
    E, K = edge_attr.shape
    if N != plan.N or E != plan.E or tuple(w_edge.shape) != (HC, K) or tuple(lin_r.weight.shape) != (HC, K_in) or \
            lin_l.weight.size(1) != K_in:
        raise ValueError("gatv2_layer_conv: operand shapes do not match the plan")
    if edge_attr.dtype != torch.float32:
        raise TypeError("gatv2_layer_conv: fp32 edge rows")
    cat_w = derived_weight("layer_conv_w", (lin_l.weight, lin_r.weight),
                           lambda: torch.cat([lin_l.weight.detach(), lin_r.weight.detach()], 0).contiguous())
    zeros = lambda m: torch.zeros(HC, dtype=torch.float32, device=dev) if m.bias is None else m.bias.detach()
    srcs = tuple(t for t in (lin_l.weight, lin_l.bias, lin_r.bias) if t is not None)
    cat_b = derived_weight("layer_conv_b", srcs, lambda: torch.cat([zeros(lin_l), zeros(lin_r)]).float().contiguous())
    wn, wn_inv = _weight_planes(cat_w, True, "f16x3")
    we, we_inv = _weight_planes(w_edge, True, "f16x3")
    (_, ntiles, cap, _), (ep, ep_inv) = plan.tiles_and_edge_planes(edge_attr, TILE_CONV_NODES, TILE_CONV_EDGES)
    tile_info = plan.tiles_heavy_first(TILE_CONV_NODES, TILE_CONV_EDGES)        # a persistent kernel: balanced rounds
    out = torch.empty(N, HC, dtype=torch.float32, device=dev)
    alpha = torch.empty(E, H, dtype=torch.float32, device=dev)
    rowmax = torch.empty(N, H, dtype=torch.float32, device=dev) if want_rowmax else None
    timer = MP_TIMER
    if timer is not None:
        ev0, ev1 = timer.bracket({"N": N, "E": E, "H": H, "C": C, "K": K, "masked": node_mask is not None or edge_mask is not None,
                                  "feat_bytes": 4, "tile_conv": True, "layer_conv": True, "K_in": K_in})
        ev0.record()
    rc = lib.isg_gatv2_layer_conv(
        x.planes.data_ptr(), x.inv.data_ptr(), wn.data_ptr(), wn_inv.data_ptr(), cat_b.data_ptr(), ep.data_ptr(), ep_inv.data_ptr(),
        we.data_ptr(), we_inv.data_ptr(), _chk(att.reshape(-1), "att", torch.float32, (HC,)),
        _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
        plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(), tile_info.data_ptr(),
        ntiles.data_ptr(), cap,
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True),
        out.data_ptr(), HC, alpha.data_ptr(), 0 if rowmax is None else rowmax.data_ptr(), N, E, H, C, K_in, K,
        float(negative_slope), _stream())
    if rc == ISG_EUNSUPPORTED:
        if timer is not None:
            timer.drop_last()
        return None
    _lib.check(rc, "isg_gatv2_layer_conv")
    if timer is not None:
        ev1.record()
    sub = _mixed_sub(plan)
    if sub is not None:
        # the gated rows of those graphs out of the SAME planes the tile kernel read (hi + mid, exact), projected per node
        pl = x.planes.view(torch.float16).index_select(0, sub.nodes).float()
        xs = ((pl[:, 0] + pl[:, 1]) * x.inv.index_select(0, sub.nodes)[:, None]).contiguous()
        y = linear(xs, cat_w, cat_b)
        _oversize_conv(sub, y[:, :HC], y[:, HC:], edge_attr, w_edge, att, H, bias, node_mask, edge_mask, negative_slope,
                       out, alpha, rowmax)
    if rowmax is not None:
        attach_row_maxima(out, rowmax)
    return out, alpha


def gatv2_tile_conv(x_l: Tensor, x_r: Tensor, edge_attr: Tensor, w_edge: Tensor, att: Tensor, plan: "GraphPlan", heads: int,
                    bias: Optional[Tensor] = None, node_mask: Optional[Tensor] = None, edge_mask: Optional[Tensor] = None,
                    negative_slope: float = 0.2, want_rowmax: bool = False):
    """gatv2_mp(x_l, x_r, lin_edge(edge_attr), ...) as ONE launch per layer (mgat_v2_conv.py:243-279 with :259-261 inside):
    isg_gatv2_tile_conv.  Bit-identical to gatv2_mp_edge_logits (the two-launch pair it replaces).  Returns (out, alpha), or
    None when the kernel has no launch for this shape (the caller then runs the pair)."""
    lib = _lib.load()
    plan.require_csr()
    N, HC = x_l.shape
    H = int(heads)
    C = HC // H
    E, K = edge_attr.shape
    if N != plan.N or E != plan.E or tuple(w_edge.shape) != (HC, K) or tuple(x_r.shape) != (N, HC):
        raise ValueError("gatv2_tile_conv: operand shapes do not match the plan")
    if x_l.dtype != torch.float32 or x_r.dtype != torch.float32 or edge_attr.dtype != torch.float32:
        raise TypeError("gatv2_tile_conv: fp32 rows")
    planes, inv = _weight_planes(w_edge, True, "f16x3")
    ep, ep_inv = plan.edge_planes(edge_attr)
    _, ntiles, cap, _ = plan.tiles(TILE_CONV_NODES, TILE_CONV_EDGES)
    tile_info = plan.tiles_heavy_first(TILE_CONV_NODES, TILE_CONV_EDGES)        # a persistent kernel: balanced rounds
    out = torch.empty(N, HC, dtype=torch.float32, device=x_l.device)
    alpha = torch.empty(E, H, dtype=torch.float32, device=x_l.device)
    rowmax = torch.empty(N, H, dtype=torch.float32, device=x_l.device) if want_rowmax else None
    timer = MP_TIMER
    if timer is not None:
        ev0, ev1 = timer.bracket({"N": N, "E": E, "H": H, "C": C, "K": K, "masked": node_mask is not None or edge_mask is not None,
                                  "feat_bytes": 4, "tile_conv": True})
        ev0.record()
    rc = lib.isg_gatv2_tile_conv(
        _chk_rows(x_l, "x_l"), x_l.stride(0), _chk_rows(x_r, "x_r"), x_r.stride(0), ep.data_ptr(), ep_inv.data_ptr(),
        planes.data_ptr(), inv.data_ptr(), _chk(att.reshape(-1), "att", torch.float32, (HC,)),
        _chk(None if bias is None else bias.reshape(-1), "bias", torch.float32, (HC,), optional=True),
        plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(), plan.dst.data_ptr(), tile_info.data_ptr(),
        ntiles.data_ptr(), cap,
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True),
        out.data_ptr(), HC, alpha.data_ptr(), 0 if rowmax is None else rowmax.data_ptr(), N, E, H, C, K,
        float(negative_slope), _stream())
    if rc == ISG_EUNSUPPORTED:
        if timer is not None:
            timer.drop_last()
        return None
    _lib.check(rc, "isg_gatv2_tile_conv")
    if timer is not None:
        ev1.record()
    sub = _mixed_sub(plan)
    if sub is not None:
        _oversize_conv(sub, x_l.index_select(0, sub.nodes), x_r.index_select(0, sub.nodes), edge_attr, w_edge, att, H, bias,
                       node_mask, edge_mask, negative_slope, out, alpha, rowmax)
    if rowmax is not None:
        attach_row_maxima(out, rowmax)
    return out, alpha



def gatv2_mp_backward(x_l: Tensor, x_r: Tensor, e_proj: Tensor, att: Tensor, alpha: Tensor, grad_out: Tensor,
                      plan: GraphPlan, heads: int, node_mask: Optional[Tensor] = None,
                      edge_mask: Optional[Tensor] = None, negative_slope: float = 0.2, want_mask_grad: bool = False):
    """Backward of gatv2_mp: (d_x_l, d_x_r, d_e_proj, d_att[H*C], d_bias[H*C], d_edge_mask[E] or None).

    No reference counterpart file: the reference relies on autograd through PyG's propagate
    (mgat_v2_conv.py:215-279); the formulas are derived in csrc/isg_mp_bwd.hip.
    """
    lib = _lib.load()
    plan.require_csr()
    N, HC = x_l.shape
    H = int(heads)
    C = HC // H
    E = plan.E
    dev = x_l.device
    x_l, x_r, e_proj, grad_out = x_l.contiguous(), x_r.contiguous(), e_proj.contiguous(), grad_out.contiguous()
    rowptr_s, eid_s, dst_s = plan.source_csr()
    d_x_l = torch.empty(N, HC, dtype=torch.float32, device=dev)
    d_x_r = torch.empty(N, HC, dtype=torch.float32, device=dev)
    d_e = torch.empty(E, HC, dtype=torch.float32, device=dev)
    blocks = (N + 15) // 16
    part = torch.empty(blocks, HC, dtype=torch.float32, device=dev)
    d_m = torch.empty(E, dtype=torch.float32, device=dev) if want_mask_grad else None
    _lib.check(lib.isg_gatv2_mp_bwd(
        _chk(x_l, "x_l", torch.float32, (N, HC)), _chk(x_r, "x_r", torch.float32, (N, HC)),
        _chk(e_proj, "e_proj", torch.float32, (E, HC)) if E > 0 else 0,
        _chk(att.reshape(-1), "att", torch.float32, (HC,)), _chk(alpha, "alpha", torch.float32, (E, H)) if E > 0 else 0,
        _chk(grad_out, "grad_out", torch.float32, (N, HC)),
        plan.rowptr.data_ptr(), plan.eid.data_ptr(), plan.src.data_ptr(),
        rowptr_s.data_ptr(), eid_s.data_ptr(), dst_s.data_ptr(),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        _chk(None if edge_mask is None else edge_mask.reshape(-1), "edge_mask", torch.float32, (E,), optional=True),
        d_x_l.data_ptr(), d_x_r.data_ptr(), d_e.data_ptr(), part.data_ptr(), 0 if d_m is None else d_m.data_ptr(),
        N, E, H, C, float(negative_slope), _stream()), "isg_gatv2_mp_bwd")
    return d_x_l, d_x_r, d_e, part.sum(0), grad_out.sum(0), d_m


def node_to_edge_mask_backward(d_edge_mask: Tensor, plan: GraphPlan) -> Tensor:
    """NodeMaskToEdgeMask.backward (sampling/node_edge_masks.py:13-19): scatter to the destination only."""
    lib = _lib.load()
    plan.require_csr()
    out = torch.empty(plan.N, dtype=torch.float32, device=d_edge_mask.device)
    _lib.check(lib.isg_node_to_edge_mask_bwd(_chk(d_edge_mask.reshape(-1), "d_edge_mask", torch.float32, (plan.E,)),
                                             plan.rowptr.data_ptr(), plan.eid.data_ptr(), out.data_ptr(), plan.N,
                                             _stream()), "isg_node_to_edge_mask_bwd")
    return out


def layer_tail_backward(ins, c, h, plan: GraphPlan, weight, bias, mean_scale, eps, node_mask, grad_out, want_mask: bool):
    """(d_ins, d_c, d_h, d_weight, d_bias, d_mean_scale, d_mask|None) of mgat_layer_tail."""
    lib = _lib.load()
    N, C = c.shape
    dev = c.device
    d_ins = torch.empty(plan.B, C, dtype=torch.float32, device=dev)
    d_c, d_h = torch.empty(N, C, dtype=torch.float32, device=dev), torch.empty(N, C, dtype=torch.float32, device=dev)
    d_mask = torch.empty(N, dtype=torch.float32, device=dev) if want_mask else None
    part = torch.empty(plan.B, 3, C, dtype=torch.float32, device=dev)
    _lib.check(lib.isg_instr_attn_graphnorm_residual_bwd(
        _chk(ins, "ins", torch.float32, (plan.B, C)), _chk(c, "c", torch.float32, (plan.N, C)),
        _chk(h, "h", torch.float32, (plan.N, C)), plan.ptr.data_ptr(), _chk(weight, "weight", torch.float32, (C,)),
        _chk(bias, "bias", torch.float32, (C,)), _chk(mean_scale, "mean_scale", torch.float32, (C,)), float(eps),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        _chk(grad_out, "grad_out", torch.float32, (N, C)), d_ins.data_ptr(), d_c.data_ptr(), d_h.data_ptr(),
        0 if d_mask is None else d_mask.data_ptr(), part.data_ptr(), plan.B, C, _stream()),
        "isg_instr_attn_graphnorm_residual_bwd")
    sums = part.sum(0)
    return d_ins, d_c, d_h, sums[0], sums[1], sums[2], d_mask


def global_attn_pool_backward(xn, q, plan: GraphPlan, node_mask, grad_out, grad_gate, want_mask: bool):
    """(d_xn, d_q, d_mask|None) of global_attn_pool."""
    lib = _lib.load()
    N, C = xn.shape
    d_xn = torch.empty(N, C, dtype=torch.float32, device=xn.device)
    d_q = torch.empty(plan.B, C, dtype=torch.float32, device=xn.device)
    d_mask = torch.empty(N, dtype=torch.float32, device=xn.device) if want_mask else None
    _lib.check(lib.isg_global_attn_pool_bwd(
        _chk(xn, "xn", torch.float32, (plan.N, C)), _chk(q, "q", torch.float32, (plan.B, C)), plan.ptr.data_ptr(),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        _chk(grad_out, "grad_out", torch.float32, (plan.B, C)),
        _chk(None if grad_gate is None else grad_gate.reshape(-1), "grad_gate", torch.float32, (N,), optional=True),
        d_xn.data_ptr(), d_q.data_ptr(), 0 if d_mask is None else d_mask.data_ptr(), plan.B, C, _stream()),
        "isg_global_attn_pool_bwd")
    return d_xn, d_q, d_mask


def instr_gate_backward(x, instr, plan: GraphPlan, grad_out):
    lib = _lib.load()
    N, C = x.shape
    d_x = torch.empty_like(x)
    d_instr = torch.empty(plan.B, C, dtype=torch.float32, device=x.device)
    _lib.check(lib.isg_instr_gate_bwd(_chk(x, "x", torch.float32, (plan.N, C)),
                                      _chk(instr, "instr", torch.float32, (plan.B, C)), plan.ptr.data_ptr(),
                                      _chk(grad_out, "grad_out", torch.float32, (N, C)), d_x.data_ptr(),
                                      d_instr.data_ptr(), plan.B, C, _stream()), "isg_instr_gate_bwd")
    return d_x, d_instr


def node_gate_backward(xn, q, batch, double_index: bool, plan: GraphPlan, grad_gate):
    """(d_xn, d_q): the per-graph partial rows are scattered into d_q here (several graphs may share a row of q)."""
    lib = _lib.load()
    N, C = xn.shape
    d_xn = torch.empty_like(xn)
    part = torch.empty(plan.B, C, dtype=torch.float32, device=xn.device)
    _lib.check(lib.isg_node_gate_bwd(_chk(xn, "xn", torch.float32, (plan.N, C)), _chk(q, "q", torch.float32, (q.size(0), C)),
                                     _chk(batch, "batch", torch.int64, (N,)), 1 if double_index else 0,
                                     plan.ptr.data_ptr(), _chk(grad_gate.reshape(-1), "grad_gate", torch.float32, (N,)),
                                     d_xn.data_ptr(), part.data_ptr(), N, plan.B, C, _stream()), "isg_node_gate_bwd")
    g = torch.arange(plan.B, device=xn.device)
    rows = batch[g.clamp(max=max(N - 1, 0))] if double_index else g
    return d_xn, torch.zeros_like(q).index_add_(0, rows, part)


def mp_algorithmic_bytes(N: int, E: int, H: int, C: int, masked: bool, feat_bytes: int = 4) -> int:
    """bytes_mp = s*(3*N*HC + E*HC) + 4*E*H + 16*E (+4*E if edge-masked)."""
    HC = H * C
    return feat_bytes * (3 * N * HC + E * HC) + 4 * E * H + 16 * E + (4 * E if masked else 0)


def edge_logits_algorithmic_bytes(N: int, E: int, H: int, C: int, K: int, masked: bool, K2: int = 0, feat_bytes: int = 4) -> int:
    """isg_gatv2_edge_logits' OWN minimum traffic: edge_attr rows (4*E*K), every x_l and x_r row once (2 * 4*N*HC), the
    logits (4*E*H), eid / src / dst (12*E), the edge mask if any (4*E).  The W planes (4*HC*K) stay in L2.  K2 > 0: the
    form that computes x_r itself reads every layer-input row once (4*N*K2) instead of x_r (4*N*HC)."""
    xr = 4 * N * K2 if K2 else feat_bytes * N * H * C
    return 4 * E * K + feat_bytes * N * H * C + xr + 4 * E * H + 12 * E + (4 * E if masked else 0)


def mp_logits_algorithmic_bytes(N: int, E: int, H: int, C: int, masked: bool, feat_bytes: int = 4) -> int:
    """isg_gatv2_mp_fwd_logits' OWN minimum traffic: x_l in and out back (2 * 4*N*HC), logits in and alpha out (8*E*H),
    the CSR (16*E as in bytes_mp), the edge mask if any (4*E)."""
    return 2 * feat_bytes * N * H * C + 8 * E * H + 16 * E + (4 * E if masked else 0)


def scatter_mean(msg: Tensor, plan: GraphPlan) -> Tensor:
    """scatter_mean(msg, dst, dim_size=N)   (scene_graph_encoder.py:141)"""
    if _rec(msg):
        from . import autograd
        return autograd.scatter_mean(msg, plan)
    lib = _lib.load()
    plan.require_csr()
    E, C = msg.shape
    if E != plan.E:
        raise ValueError(f"msg has {E} rows, plan has {plan.E} edges")
    out = torch.empty(plan.N, C, dtype=torch.float32, device=msg.device)
    _lib.check(lib.isg_scatter_mean(_chk(msg, "msg", torch.float32), plan.rowptr.data_ptr(), plan.eid.data_ptr(),
                                    out.data_ptr(), plan.N, C, _stream()), "isg_scatter_mean")
    return out


# ------------------------------------------------------------------------------------------------
# Node gate + samplers
# ------------------------------------------------------------------------------------------------
def node_gate(xn: Tensor, q: Tensor, batch: Tensor, double_index: bool, plan: Optional["GraphPlan"] = None) -> Tensor:
    """gelu(<xn_n, q[r(n)]>/sqrt(C)) -> [N,1]   (masking.py:151-155).  ``plan`` selects the HIP backward."""
    if _rec(xn, q):
        from . import autograd
        return autograd.node_gate(xn, q, batch, double_index, plan)
    lib = _lib.load()
    N, C = xn.shape
    gate = torch.empty(N, 1, dtype=torch.float32, device=xn.device)
    _lib.check(lib.isg_node_gate(_chk(xn, "xn", torch.float32), _chk(q, "q", torch.float32, (q.size(0), C)),
                                 _chk(batch, "batch", torch.int64, (N,)), 1 if double_index else 0, gate.data_ptr(),
                                 N, C, _stream()), "isg_node_gate")
    return gate


# CFG.fuse_gate (ops.FUSE_GATE): the masked layer's node gate from the layer input's planes, node_nn inside (A/B switch)


def node_gate_planes_supported(node_nn: torch.nn.Sequential, q: Tensor) -> bool:
    """Shape test of isg_node_gate_planes: inference, node_nn = Linear(128 -> 128) + exact GELU, 128-wide question rows."""
    if not (CFG.fuse_gate and CFG.gemm_backend == "bf16x6" and CFG.gemm_f16x3) or torch.is_grad_enabled():
        return False
    mods = list(node_nn)
    if len(mods) != 2 or not isinstance(mods[0], torch.nn.Linear) or not isinstance(mods[1], torch.nn.GELU) \
            or mods[1].approximate != "none" or mods[0].bias is None:
        return False
    return tuple(mods[0].weight.shape) == (128, 128) and q.dim() == 2 and q.size(1) == 128 and q.dtype == torch.float32


def node_gate_planes(x: "NodePlanes", node_nn: torch.nn.Sequential, q: Tensor, batch: Tensor, double_index: bool) -> Tensor:
    """gelu(<gelu(node_nn(x))_n, q[r(n)]>/sqrt(C)) -> [N,1] (masking.py:137, 151-155) from the layer input as NodePlanes:
    isg_node_gate_planes.  The caller checks node_gate_planes_supported() first."""
    lib = _lib.load()
    N = batch.numel()
    l0 = node_nn[0]
    wp, w_inv = _weight_planes(l0.weight, True, "f16x3")
    gate = torch.empty(N, 1, dtype=torch.float32, device=q.device)
    _lib.check(lib.isg_node_gate_planes(x.planes.data_ptr(), x.inv.data_ptr(), wp.data_ptr(), w_inv.data_ptr(),
                                        _chk(l0.bias.detach(), "node_nn.0.bias", torch.float32, (128,)),
                                        _chk(q, "q", torch.float32, (q.size(0), 128)), _chk(batch, "batch", torch.int64, (N,)),
                                        1 if double_index else 0, gate.data_ptr(), N, 128, _stream()), "isg_node_gate_planes")
    return gate


def _rows(scores: Tensor, plan: Optional[GraphPlan]):
    """(ptr, B, nmax_host, nmax_dev, out) for the ragged (plan) or dense ([B,Nmax]) row layout."""
    if plan is not None:
        flat = scores.reshape(-1)
        if flat.numel() != plan.N:
            raise ValueError(f"scores has {flat.numel()} entries, plan has {plan.N} nodes")
        return flat, plan.ptr.data_ptr(), plan.B, plan.nmax, plan.nmax_dev.data_ptr()
    if scores.dim() != 2:
        raise ValueError("dense scores must be [B, Nmax]")
    if scores.size(1) > MAX_NODES_PER_GRAPH:
        raise _lib.IsgError(f"rows longer than {MAX_NODES_PER_GRAPH} slots are unsupported")
    return scores, 0, scores.size(0), scores.size(1), 0


def _gid_ptr(plan: Optional["GraphPlan"]) -> int:
    """The graph-id table of a plan that is a CUT of a larger batch (run_split's sub-batch: plan.graph_ids = int32 [B], the graphs'
    numbers in the batch they came from): row b of a sampler then draws the Philox stream of graph graph_ids[b], i.e. the noise
    that graph gets when the batch is not split (seeded masks do not depend on the dispatch).  0: row b draws stream b."""
    g = None if plan is None else plan.graph_ids
    if g is None:
        return 0
    if g.dtype != torch.int32 or g.numel() != plan.B or not g.is_contiguous():
        raise ValueError("plan.graph_ids must be a contiguous int32 [B] tensor")
    return g.data_ptr()


def _noise_ptr(noise: Optional[Tensor], B: int, nmax: int) -> int:
    if noise is None:
        return 0
    if noise.numel() != B * nmax:
        raise ValueError(f"noise must hold B*Nmax = {B}*{nmax} values, got {tuple(noise.shape)}")
    return _chk(noise.reshape(B, nmax), "noise", torch.float32)


def topk_gumbel(scores: Tensor, k: int, tau: float = 0.1, plan: Optional[GraphPlan] = None,
                noise: Optional[Tensor] = None, seed: int = 0, return_khot: bool = False):
    """Relaxed Gumbel top-k + straight-through hard mask (gumbel_scheme.py:55-104).

    Ragged (plan given): scores [N] / [N,1] -> mask of the same shape (the dense-pad and the `[mask]`
    un-pad of masking.py:162,176 are fused).  Dense: scores [B,Nmax] -> [B,Nmax].
    """
    if _rec(scores):
        from . import autograd
        if return_khot:
            raise ValueError("return_khot is an inspection output and is not differentiable")
        return autograd.topk_gumbel(scores, k, tau, plan, noise, seed)
    lib = _lib.load()
    flat, ptr, B, nmax, nmax_dev = _rows(scores, plan)
    out = torch.empty_like(flat)
    khot = torch.empty(B, nmax, dtype=torch.float32, device=scores.device) if return_khot else None
    _lib.check(lib.isg_topk_gumbel(_chk(flat, "scores", torch.float32), ptr, B, nmax, nmax_dev,
                                   _noise_ptr(noise, B, nmax), int(seed) & (2 ** 64 - 1), _gid_ptr(plan), int(k), float(tau),
                                   out.data_ptr(), 0 if khot is None else khot.data_ptr(), _stream()),
               "isg_topk_gumbel")
    out = out.view(scores.shape)
    return (out, khot) if return_khot else out


def topk_threshold(scores: Tensor, k: int, plan: Optional[GraphPlan] = None, noise: Optional[Tensor] = None,
                   noise_scale: float = 0.0, seed: int = 0, return_dense: bool = False):
    """(scores + noise*noise_scale) >= k-th largest, per row (deterministic_scheme.py:36-43).  Not differentiable by
    itself: the I-MLE / AIMLE estimators around it are autograd.imle_topk / aimle_topk.  ``return_dense`` also returns
    the selection over the padded [B, Nmax] rows (pads included)."""
    lib = _lib.load()
    scores = scores.detach()
    flat, ptr, B, nmax, nmax_dev = _rows(scores, plan)
    out = torch.empty_like(flat)
    dense = torch.empty(B, nmax, dtype=torch.float32, device=scores.device) if return_dense else None
    _lib.check(lib.isg_topk_threshold(_chk(flat, "scores", torch.float32), ptr, B, nmax, nmax_dev,
                                      _noise_ptr(noise, B, nmax), float(noise_scale), int(seed) & (2 ** 64 - 1), _gid_ptr(plan),
                                      int(k), out.data_ptr(), 0 if dense is None else dense.data_ptr(), _stream()),
               "isg_topk_threshold")
    out = out.view(scores.shape)
    return (out, dense) if return_dense else out


def simple_topk(scores: Tensor, k: int, plan: Optional[GraphPlan] = None, uniform: Optional[Tensor] = None,
                seed: int = 0, return_marginals: bool = False):
    """SIMPLE sampler (simple_scheme.py:44-162): mask = (Gumbel top-k sample - marginals) + marginals, marginals exact
    through the exactly-k circuit.  Ragged (plan given; plan.nmax must be the batch's true longest row): scores [N] /
    [N,1]; dense: scores [B,Nmax].  ``uniform``: the [B, n] torch.rand draw (n = Nmax rounded up to a power of two)."""
    if _rec(scores):
        from . import autograd
        return autograd.simple_topk(scores, k, plan, uniform, seed, return_marginals)
    lib = _lib.load()
    flat, ptr, B, nmax, _ = _rows(scores, plan)
    out = torch.empty_like(flat)
    n = 1 << max(nmax - 1, 0).bit_length() if nmax > 0 else 0
    if uniform is not None and uniform.numel() != B * n:
        raise ValueError(f"uniform must hold B*n = {B}*{n} values, got {tuple(uniform.shape)}")
    marg = torch.empty(B, nmax, dtype=torch.float32, device=scores.device) if return_marginals else None
    _lib.check(lib.isg_simple_topk(_chk(flat, "scores", torch.float32), ptr, B, nmax,
                                   0 if uniform is None else _chk(uniform.reshape(B, n), "uniform", torch.float32),
                                   int(seed) & (2 ** 64 - 1), _gid_ptr(plan), int(k), out.data_ptr(),
                                   0 if marg is None else marg.data_ptr(), _stream()), "isg_simple_topk")
    out = out.view(scores.shape)
    return (out, marg) if return_marginals else out


def topk_gumbel_backward(scores: Tensor, grad_out: Tensor, k: int, tau: float = 0.1, plan: Optional[GraphPlan] = None,
                         noise: Optional[Tensor] = None, seed: int = 0) -> Tensor:
    """d scores of topk_gumbel for d out = grad_out (straight-through, gumbel_scheme.py:83-90); same arguments as the
    forward (the relaxation is replayed from scores + noise/seed)."""
    lib = _lib.load()
    flat, ptr, B, nmax, nmax_dev = _rows(scores, plan)
    g = grad_out.reshape(flat.shape).contiguous()
    out = torch.empty_like(flat)
    _lib.check(lib.isg_topk_gumbel_bwd(_chk(flat, "scores", torch.float32), ptr, B, nmax, nmax_dev,
                                       _noise_ptr(noise, B, nmax), int(seed) & (2 ** 64 - 1), _gid_ptr(plan), int(k), float(tau),
                                       _chk(g, "grad_out", torch.float32), out.data_ptr(), _stream()),
               "isg_topk_gumbel_bwd")
    return out.view(scores.shape)


# ------------------------------------------------------------------------------------------------
# Per-graph attention / norm / pooling
# ------------------------------------------------------------------------------------------------
def scatter_attention(query: Tensor, key: Tensor, plan: GraphPlan, value: Optional[Tensor] = None) -> Tensor:
    """softmax_g(<query_g, key_n>/sqrt(C)) * value_n   (utils/scatter_scaled_dot_product.py:6-15)"""
    if value is None:
        value = key
    if _rec(query, key, value):
        from . import autograd
        return autograd.scatter_attention(query, key, plan, value)
    lib = _lib.load()
    N, C = key.shape
    out = torch.empty_like(value)
    _lib.check(lib.isg_scatter_attention(_chk(query, "query", torch.float32, (plan.B, C)),
                                         _chk(key, "key", torch.float32, (plan.N, C)),
                                         _chk(value, "value", torch.float32, (plan.N, C)),
                                         plan.ptr.data_ptr(), out.data_ptr(), plan.B, C, _stream()),
               "isg_scatter_attention")
    return out


def graph_norm(x: Tensor, plan: GraphPlan, weight: Tensor, bias: Tensor, mean_scale: Tensor, eps: float = 1e-5,
               fp64: bool = False) -> Tensor:
    """PyG GraphNorm forward (mgat.py:171); fp64=True mirrors scene_graph_encoder.py:99-102."""
    if _rec(x, weight, bias, mean_scale):
        from . import autograd
        return autograd.graph_norm(x, plan, weight, bias, mean_scale, eps, fp64)
    lib = _lib.load()
    N, C = x.shape
    out = torch.empty_like(x)
    _lib.check(lib.isg_graph_norm(_chk(x, "x", torch.float32, (plan.N, C)), plan.ptr.data_ptr(),
                                  _chk(weight, "weight", torch.float32, (C,)), _chk(bias, "bias", torch.float32, (C,)),
                                  _chk(mean_scale, "mean_scale", torch.float32, (C,)), float(eps), 1 if fp64 else 0,
                                  out.data_ptr(), plan.B, C, _stream()), "isg_graph_norm")
    return out


def mgat_layer_tail(ins: Tensor, c: Tensor, h: Tensor, plan: GraphPlan, weight: Tensor, bias: Tensor,
                    mean_scale: Tensor, eps: float = 1e-5, node_mask: Optional[Tensor] = None) -> Tensor:
    """scatter attention -> GraphNorm -> + h [-> * mask], fused (mgat.py:168-177)."""
    if _rec(ins, c, h, weight, bias, mean_scale, node_mask):
        from . import autograd
        return autograd.mgat_layer_tail(ins, c, h, plan, weight, bias, mean_scale, eps, node_mask)
    lib = _lib.load()
    N, C = c.shape
    out = torch.empty_like(h)
    _lib.check(lib.isg_instr_attn_graphnorm_residual(
        _chk(ins, "ins", torch.float32, (plan.B, C)), _chk(c, "c", torch.float32, (plan.N, C)),
        _chk(h, "h", torch.float32, (plan.N, C)), plan.ptr.data_ptr(), _chk(weight, "weight", torch.float32, (C,)),
        _chk(bias, "bias", torch.float32, (C,)), _chk(mean_scale, "mean_scale", torch.float32, (C,)), float(eps),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        out.data_ptr(), plan.B, C, _stream()), "isg_instr_attn_graphnorm_residual")
    return out


# CFG.fuse_dense_tail (ops.FUSE_DENSE_TAIL): x_proj + layer tail + next instruction gate as one kernel on graph-aligned tiles (A/B switch)
# CFG.dense_tail_rows (ops.DENSE_TAIL_ROWS): nodes per tile of isg_mgat_dense_tail


def dense_tail_supported(plan: GraphPlan, x_proj: torch.nn.Sequential, width_in: int, channels: int) -> bool:
    """Shape test of isg_mgat_dense_tail (csrc/isg_layer_tile.hip): inference, fp32, Linear(512 -> 256) GELU Linear(256 ->
    128) GELU (MGAT at C = 128, H = 4: BASELINE configs[1]), every graph within one 64-node tile."""
    if not (CFG.fuse_dense_tail and CFG.gemm_backend == "bf16x6" and CFG.gemm_f16x3) or torch.is_grad_enabled():
        return False
    mods = list(x_proj)
    if len(mods) != 4 or not all(isinstance(m, torch.nn.GELU) and m.approximate == "none" for m in (mods[1], mods[3])):
        return False
    l0, l2 = mods[0], mods[2]
    if not (isinstance(l0, torch.nn.Linear) and isinstance(l2, torch.nn.Linear)) or l0.bias is None or l2.bias is None:
        return False
    return (width_in == 512 and channels == 128 and tuple(l0.weight.shape) == (256, 512) and
            tuple(l2.weight.shape) == (128, 256) and plan.B > 0 and plan.batch is not None and
            plan.tile_mode(CFG.dense_tail_rows, TILE_CONV_EDGES) != "none")


def mgat_dense_tail(conv_out: Tensor, x_proj: torch.nn.Sequential, ins: Tensor, h: Tensor, plan: GraphPlan, weight: Tensor,
                    bias: Tensor, mean_scale: Tensor, eps: float = 1e-5, node_mask: Optional[Tensor] = None,
                    ins_next: Optional[Tensor] = None, want_rows: bool = True, want_planes: bool = False
                    ) -> Optional[Tuple[Tensor, Optional[Tensor], Optional["NodePlanes"]]]:
    """mgat.py:156-177 after the convolution, plus the next layer's instruction gate (mgat_v2_conv.py:156-157), as one launch:
    x_proj (Linear GELU Linear GELU) -> scatter attention -> GraphNorm -> + h [-> * mask] -> (h', xg, xg planes) with
    xg = gelu(h' * ins_next[batch]) as fp32 rows (want_rows) and / or as NodePlanes for gatv2_layer_conv (want_planes).
    conv_out must carry its row maxima (the message-passing kernels leave them); returns None when it does not (the caller
    runs the un-fused chain).  The caller checks dense_tail_supported() first."""
    lib = _lib.load()
    N, K1 = conv_out.shape
    rm = row_maxima(conv_out)
    if rm is None or rm.dim() != 2 or rm.size(0) != N or rm.stride(1) != 1 or rm.dtype != torch.float32:
        return None
    l0, l2 = x_proj[0], x_proj[2]
    C = l2.weight.size(0)
    p1, inv1 = _weight_planes(l0.weight, True, "f16x3")
    p2, inv2 = _weight_planes(l2.weight, True, "f16x3")
    ybound = derived_weight("dense_tail_bound", (l0.weight, l0.bias), lambda: torch.stack(
        [l0.weight.detach().abs().sum(dim=1).max(), l0.bias.detach().abs().max()]).float().contiguous())
    # one tile plan per batch: the convolution's (64 nodes / 256 slots) serves this kernel too when it exists
    # one tile plan per batch, the convolution's (64 nodes / 256 slots): the same graphs are "oversize" for every tile kernel
    tile_ptr, ntiles, cap, tile_info = plan.tiles(CFG.dense_tail_rows, TILE_CONV_EDGES if plan.rowptr is not None else 0)
    h_out = torch.empty_like(h)
    xg = torch.empty_like(h) if ins_next is not None and want_rows else None
    xp = None
    if ins_next is not None and want_planes:
        xp = NodePlanes(torch.empty(max(N, 1), 2, 128, dtype=torch.int16, device=h.device),
                        torch.empty(max(N, 1), dtype=torch.float32, device=h.device))
    rc = lib.isg_mgat_dense_tail(
        _chk_rows(conv_out, "conv_out"), conv_out.stride(0), rm.data_ptr(), rm.size(1), rm.stride(0),
        p1.data_ptr(), inv1.data_ptr(), _chk(l0.bias.detach(), "x_proj.0.bias", torch.float32, (l0.weight.size(0),)),
        ybound.data_ptr(), p2.data_ptr(), inv2.data_ptr(), _chk(l2.bias.detach(), "x_proj.2.bias", torch.float32, (C,)),
        _chk(ins, "ins", torch.float32, (plan.B, C)), _chk(h, "h", torch.float32, (plan.N, C)),
        _chk(weight.detach(), "weight", torch.float32, (C,)), _chk(bias.detach(), "bias", torch.float32, (C,)),
        _chk(mean_scale.detach(), "mean_scale", torch.float32, (C,)), float(eps),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        _chk(ins_next, "ins_next", torch.float32, (plan.B, C), optional=True), h_out.data_ptr(),
        0 if xg is None else xg.data_ptr(), 0 if xp is None else xp.planes.data_ptr(), 0 if xp is None else xp.inv.data_ptr(),
        plan.ptr.data_ptr(), _chk(plan.batch, "batch", torch.int64, (N,)),
        tile_ptr.data_ptr(), tile_info.data_ptr(), ntiles.data_ptr(), cap, N, K1, l0.weight.size(0), C, _stream())
    if rc == ISG_EUNSUPPORTED:
        return None
    _lib.check(rc, "isg_mgat_dense_tail")
    sub = _mixed_sub(plan)
    if sub is not None:
        # mgat.py:156-177 on the graphs the tile kernel passed over: x_proj, the per-graph layer tail, the next gate
        cs = mlp(x_proj, conv_out.index_select(0, sub.nodes))
        nm = None if node_mask is None else node_mask.reshape(-1).index_select(0, sub.nodes).view(-1, 1)
        hs = mgat_layer_tail(ins.index_select(0, sub.gids), cs.contiguous(), h.index_select(0, sub.nodes), sub.plan, weight, bias,
                             mean_scale, eps, node_mask=nm)
        h_out.index_copy_(0, sub.nodes, hs)
        if ins_next is not None and (xg is not None or xp is not None):
            rows, pl = instr_gate_planes(hs, ins_next.index_select(0, sub.gids), sub.batch, want_rows=xg is not None)
            if xg is not None:
                xg.index_copy_(0, sub.nodes, rows)
            if xp is not None:
                xp.planes.index_copy_(0, sub.nodes, pl.planes[:sub.nodes.numel()])
                xp.inv.index_copy_(0, sub.nodes, pl.inv[:sub.nodes.numel()])
    return h_out, xg, xp


# CFG.fuse_readout (ops.FUSE_READOUT): node_nn + mask + per-graph softmax pooling as one launch on graph-aligned tiles (A/B switch)


def readout_tile_supported(plan: GraphPlan, node_nn: torch.nn.Sequential, width_in: int) -> bool:
    """Shape test of isg_readout_tile: inference, fp32, node_nn = Linear(128 -> 128) GELU Linear(128 -> 128), tiles of 64 nodes."""
    if not (CFG.fuse_readout and CFG.gemm_backend == "bf16x6" and CFG.gemm_f16x3) or torch.is_grad_enabled():
        return False
    mods = list(node_nn)
    if len(mods) != 3 or not isinstance(mods[1], torch.nn.GELU) or mods[1].approximate != "none":
        return False
    l0, l2 = mods[0], mods[2]
    if not (isinstance(l0, torch.nn.Linear) and isinstance(l2, torch.nn.Linear)) or l0.bias is None or l2.bias is None:
        return False
    return (width_in == 128 and tuple(l0.weight.shape) == (128, 128) and tuple(l2.weight.shape) == (128, 128) and plan.B > 0 and
            plan.batch is not None and plan.tile_mode(CFG.dense_tail_rows, TILE_CONV_EDGES) != "none")


def readout_tile(x: Tensor, node_nn: torch.nn.Sequential, q: Tensor, plan: GraphPlan, node_mask: Optional[Tensor] = None):
    """GlobalAttention.forward's device work as ONE launch (att_pooling.py:57-77): node_nn(x) * mask, per-graph softmax of
    <xn, q>/sqrt(C), pooled sum.  Returns (out [B,C], gate [N,1]) or None when the kernel has no launch for this shape."""
    lib = _lib.load()
    N, C = x.shape
    l0, l2 = node_nn[0], node_nn[2]
    p1, inv1 = _weight_planes(l0.weight, True, "f16x3")
    p2, inv2 = _weight_planes(l2.weight, True, "f16x3")
    ybound = derived_weight("dense_tail_bound", (l0.weight, l0.bias), lambda: torch.stack(
        [l0.weight.detach().abs().sum(dim=1).max(), l0.bias.detach().abs().max()]).float().contiguous())
    tile_ptr, ntiles, cap, tile_info = plan.tiles(CFG.dense_tail_rows, TILE_CONV_EDGES if plan.rowptr is not None else 0)
    out = torch.empty(plan.B, C, dtype=torch.float32, device=x.device)
    gate = torch.empty(N, 1, dtype=torch.float32, device=x.device)
    rc = lib.isg_readout_tile(
        _chk_rows(x, "x"), x.stride(0), p1.data_ptr(), inv1.data_ptr(), _chk(l0.bias.detach(), "node_nn.0.bias", torch.float32, (C,)),
        ybound.data_ptr(), p2.data_ptr(), inv2.data_ptr(), _chk(l2.bias.detach(), "node_nn.2.bias", torch.float32, (C,)),
        _chk(q, "q", torch.float32, (plan.B, C)),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        out.data_ptr(), gate.data_ptr(), plan.ptr.data_ptr(), _chk(plan.batch, "batch", torch.int64, (N,)), tile_ptr.data_ptr(),
        tile_info.data_ptr(), ntiles.data_ptr(), cap, N, C, _stream())
    if rc == ISG_EUNSUPPORTED:
        return None
    _lib.check(rc, "isg_readout_tile")
    sub = _mixed_sub(plan)
    if sub is not None:
        # att_pooling.py:62-73 on the graphs the tile kernel passed over (it zeroed their rows of `out`)
        xn = mlp(node_nn, x.index_select(0, sub.nodes))
        nm = None if node_mask is None else node_mask.reshape(-1).index_select(0, sub.nodes).view(-1, 1)
        o, g = global_attn_pool(xn.contiguous(), q.index_select(0, sub.gids), sub.plan, nm)
        out.index_copy_(0, sub.gids, o)
        gate.index_copy_(0, sub.nodes, g)
    return out, gate


def global_attn_pool(xn: Tensor, q: Tensor, plan: GraphPlan, node_mask: Optional[Tensor] = None):
    """GlobalAttention soft-max pooling (att_pooling.py:63-73).  Returns (out[B,C], gate[N,1])."""
    if _rec(xn, q, node_mask):
        from . import autograd
        return autograd.global_attn_pool(xn, q, plan, node_mask)
    lib = _lib.load()
    N, C = xn.shape
    out = torch.empty(plan.B, C, dtype=torch.float32, device=xn.device)
    gate = torch.empty(N, 1, dtype=torch.float32, device=xn.device)
    _lib.check(lib.isg_global_attn_pool(
        _chk(xn, "xn", torch.float32, (plan.N, C)), _chk(q, "q", torch.float32, (plan.B, C)), plan.ptr.data_ptr(),
        _chk(None if node_mask is None else node_mask.reshape(-1), "node_mask", torch.float32, (N,), optional=True),
        out.data_ptr(), gate.data_ptr(), plan.B, C, _stream()), "isg_global_attn_pool")
    return out, gate


# ------------------------------------------------------------------------------------------------
# Dense projections: fp32 accuracy on the bf16 matrix cores (csrc/isg_gemm.hip)
# ------------------------------------------------------------------------------------------------
# CFG.gemm_backend (ops.GEMM_BACKEND): "bf16x6": this library's kernels; "torch": hipBLASLt fp32 through torch (A/B switch)
_PLANES = {}                 # (id(weight), layout) -> (weakref, version, data_ptr, planes): static weights are split once
# CFG.f16x3_f16_out (ops.F16X3_F16_OUT): half-row results (configs[4]) of K <= 128 Linears on isg_linear_f16x3_f16 instead of the bf16 six-product
                             # panel kernel (A/B switch)
# CFG.gemm_kernel (ops.GEMM_KERNEL): "auto": per shape (below); "panel": isg_linear_panel; "tile": isg_linear_bf16x6 (A/B switch)


# CFG.f16x3_tile (ops.F16X3_TILE): ... and 128 < K <= 1024 when the producer of the input left its row maxima (isg_linear_f16x3_tile)
# CFG.gemm_f16x3 (ops.GEMM_F16X3): K <= 128 panel shapes on the fp16 three-product kernel (isg_linear_f16x3) instead of bf16x6 (A/B switch)


# CFG.panel_min_n (ops.PANEL_MIN_N): narrowest Linear the row-panel kernels take (A/B: tools/ab_step.py)


def _use_panel(M: int, N: int, K: int) -> bool:
    """The row-panel kernel wins where an A panel is split once and serves many columns (K <= 128: lin_edge 182 vs 212 us,
    lin_l|lin_r 153 vs 166 us) and there are enough 64-row panels to fill the chip; the tile kernel elsewhere
    (profiles/r02_a_gemm_structures.md)."""
    if CFG.gemm_kernel != "auto":
        return CFG.gemm_kernel == "panel"
    return K <= 128 and N >= CFG.panel_min_n and M >= 32768


_DERIVED = {}   # (tag, ids of the source tensors) -> (versions, weakrefs, value): weights re-laid-out once per model


def derived_weight(tag: str, sources, build):
    """Cache of tensors computed from static weights (slices, concatenations): rebuilt when a source was updated in place
    (tensor._version / data_ptr) or replaced; invalidate_weight_cache() drops it."""
    key = (tag,) + tuple(id(t) for t in sources)
    ver = tuple((_ver(t), t.data_ptr()) for t in sources)
    hit = _DERIVED.get(key)
    if hit is not None and hit[0] == ver and all(r() is t for r, t in zip(hit[1], sources)):
        hit[3].wait()               # built on another stream a moment ago (run_split's two passes share this cache)?
        return hit[2]
    with torch.no_grad():
        value = build()
    if len(_DERIVED) > 256:
        for k in [k for k, v in _DERIVED.items() if any(r() is None for r in v[1])]:
            del _DERIVED[k]
    _DERIVED[key] = (ver, tuple(weakref.ref(t) for t in sources), value, _Ready(any(t.is_cuda for t in sources)))
    return value


def gather_add(A: Tensor, ia: Tensor, B: Optional[Tensor] = None, ib: Optional[Tensor] = None, T: Optional[Tensor] = None,
               it: Optional[Tensor] = None, sign: Optional[Tensor] = None, D: Optional[Tensor] = None,
               bias: Optional[Tensor] = None, gelu: bool = False, planes_out: bool = False):
    """act(A[ia] + B[ib] + sign * T[it] + D + bias) -> [E, C]: the per-edge remainder of a Linear over
    cat([x[row], x[col], emb]) once its node parts are projected per node (scene_graph_encoder.py:119-120,139-140;
    csrc/isg_sgenc.hip).  A / B / T / D may be column slices of wider tensors (row stride a multiple of 4).
    planes_out: the rows as Planes32 only (the operand of the Linear that follows, no split pass, no fp32 rows)."""
    lib = _lib.load()
    E, C = ia.numel(), A.size(1)
    out = None if planes_out else torch.empty(E, C, dtype=torch.float32, device=A.device)
    pl = torch.empty(int(lib.isg_planes32_elems(E, C)), dtype=torch.int16, device=A.device) if planes_out else None
    pinv = torch.empty(E, dtype=torch.float32, device=A.device) if planes_out else None

    def rows(t, name):
        if t is None:
            return 0, 0
        if tuple(t.shape[1:]) != (C,):
            raise ValueError(f"{name}: expected [*, {C}], got {tuple(t.shape)}")
        return _chk_rows(t, name), t.stride(0)

    def idx(t, name, n):
        if t is None:
            return 0
        if t.numel() != n:
            raise ValueError(f"{name}: expected {n} indices, got {t.numel()}")
        return _chk(t.reshape(-1), name, torch.int64)

    pa, la = rows(A, "A")
    pb, lb = rows(B, "B")
    pt, lt = rows(T, "T")
    pd, ld = rows(D, "D")
    if D is not None and D.size(0) != E:
        raise ValueError(f"D: expected {E} rows, got {D.size(0)}")
    _lib.check(lib.isg_gather_add(pa, idx(ia, "ia", E), la, pb, idx(ib, "ib", E), lb, pt, idx(it, "it", E),
                                  _chk(None if sign is None else sign.reshape(-1), "sign", torch.float32, (E,), optional=True),
                                  lt, pd, ld, _chk(bias, "bias", torch.float32, (C,), optional=True),
                                  0 if out is None else out.data_ptr(), E, C, 1 if gelu else 0,
                                  0 if pl is None else pl.data_ptr(), 0 if pinv is None else pinv.data_ptr(), _stream()),
               "isg_gather_add")
    return Planes32(pl, pinv, E, C) if planes_out else out


# CFG.embedding_sum (ops.EMBEDDING_SUM): sum of a node's token embeddings through isg_gather_add instead of gather + reduce (A/B switch)


def embedding_sum(weight: Tensor, idx: Tensor) -> Tensor:
    """sum_t weight[idx[:, t]] -> [N, C]: torch.sum(embedding(idx), dim=-2) (scene_graph_encoder.py:63-70) without the [N, T, C]
    intermediate -- isg_gather_add adds up to three gathered rows (and a dense term) per launch, so four tokens are two launches
    over a table that sits in L2 (1.5 MB) instead of a 79 us gather x 2 and a 116 us reduction at 82 k nodes.  Inference, fp32,
    4 | C; anything else: the torch ops.  (The sum runs ((t0 + t1) + t2) then + t3: equal to torch's to rounding.)"""
    if (not CFG.embedding_sum or _rec(weight) or weight.dtype != torch.float32 or idx.dim() != 2 or idx.size(1) < 2 or
            weight.size(1) % 4 != 0 or not weight.is_cuda or idx.dtype != torch.int64):
        return torch.sum(torch.nn.functional.embedding(idx, weight), dim=-2)
    w = weight.detach()
    cols = [idx[:, t].contiguous() for t in range(idx.size(1))]
    out, t = None, 0
    while t < len(cols):
        take = cols[t:t + 3] if out is None else cols[t:t + 2]
        args = [w, take[0]]
        args += [w, take[1]] if len(take) > 1 else [None, None]
        args += [w, take[2]] if len(take) > 2 else [None, None]
        out = gather_add(*args, None, out)
        t += len(take)
    return out


def invalidate_weight_cache() -> None:
    """Drop every cached bf16 plane set and fused weight.  The caches are validated by (object identity, tensor._version,
    data_ptr); a write THROUGH `.data` (weight.data.copy_/mul_, as init / EMA / weight-surgery code does) bumps neither,
    so such code must call this (Module.load_state_dict goes through copy_ on the Parameter and is safe)."""
    _PLANES.clear()
    _CAT.clear()
    _DERIVED.clear()


def _weight_planes(weight: Tensor, cache: bool = True, layout: str = "tile") -> Tensor:
    key = (id(weight), layout)
    hit = _PLANES.get(key) if cache else None
    # the weak reference pins the identity: a freed weight's id (and even its address) can be reused by another model
    if hit is not None and hit[0]() is weight and hit[1] == _ver(weight) and hit[2] == weight.data_ptr():
        hit[4].wait()
        return hit[3]
    lib = _lib.load()
    N, K = weight.shape
    w = weight.detach()
    if layout == "f16x3_rows":  # row-major scaled fp16 planes of isg_linear_f16x3_tile
        Kp = (K + 31) // 32 * 32
        planes = torch.empty(2 * N * Kp, dtype=torch.int16, device=weight.device)
        inv = torch.empty(N, dtype=torch.float32, device=weight.device)
        _lib.check(lib.isg_split_f16x2_rows(_chk(w.contiguous(), "weight", torch.float32), N, K, planes.data_ptr(),
                                            inv.data_ptr(), _stream()), "isg_split_f16x2_rows")
        planes = (planes, inv)
    elif layout == "f16x3":      # two scaled fp16 planes + the inverse row scales of isg_linear_f16x3
        planes = torch.empty(int(lib.isg_split_f16x2_frag_elems(N, K)), dtype=torch.int16, device=weight.device)
        inv = torch.empty((N + 31) // 32 * 32, dtype=torch.float32, device=weight.device)
        _lib.check(lib.isg_split_f16x2_frag(_chk(w.contiguous(), "weight", torch.float32), N, K, planes.data_ptr(),
                                            inv.data_ptr(), _stream()), "isg_split_f16x2_frag")
        planes = (planes, inv)
    elif layout == "panel":      # fragment-major planes of isg_linear_panel
        planes = torch.empty(int(lib.isg_split_bf16x3_frag_elems(N, K)), dtype=torch.int16, device=weight.device)
        _lib.check(lib.isg_split_bf16x3_frag(_chk(w.contiguous(), "weight", torch.float32), N, K, planes.data_ptr(),
                                             _stream()), "isg_split_bf16x3_frag")
    else:
        Kp = (K + 31) // 32 * 32
        planes = torch.empty(3 * N * Kp, dtype=torch.int16, device=weight.device)
        _lib.check(lib.isg_split_bf16x3(_chk(w.contiguous(), "weight", torch.float32), N, K, planes.data_ptr(), _stream()),
                   "isg_split_bf16x3")
    if cache:
        if len(_PLANES) > 256:
            for k in [k for k, v in _PLANES.items() if v[0]() is None]:
                del _PLANES[k]
        _PLANES[key] = (weakref.ref(weight), _ver(weight), weight.data_ptr(), planes, _Ready(weight.is_cuda))
    return planes


def attach_row_maxima(x: Tensor, rowmax: Tensor) -> Tensor:
    """Leave partial row maxima [M, P] (max |x| over P equal column blocks of every row) on `x` for the fp16 three-product
    Linears that read it.  They are tied to x's version counter: an in-place write to x afterwards invalidates them
    (`row_maxima` then returns None and the Linear makes its own pass) -- a stale maximum would mis-scale the fp16 planes."""
    x._isg_rowmax = rowmax
    x._isg_rowmax_version = (_ver(x), x.data_ptr(), tuple(x.shape))
    return x


def row_maxima(x: Tensor) -> Optional[Tensor]:
    rm = getattr(x, "_isg_rowmax", None)
    if rm is None or getattr(x, "_isg_rowmax_version", None) != (_ver(x), x.data_ptr(), tuple(x.shape)):
        return None
    return rm


def carry_row_maxima(dst: Tensor, src: Tensor) -> Tensor:
    """`dst` is a row-order-preserving view / reshape of `src` (same rows, same values): it keeps src's maxima, and src's
    planes32 (what a Linear on the engine would otherwise split again)."""
    rm = row_maxima(src)
    if rm is not None and dst.dim() >= 1:
        attach_row_maxima(dst, rm)
    hit = getattr(src, "_isg_planes32", None)
    if hit is not None and hit[0] == (_ver(src), src.data_ptr(), tuple(src.shape)) and dst.data_ptr() == src.data_ptr() \
            and dst.numel() == src.numel():
        dst._isg_planes32 = ((_ver(dst), dst.data_ptr(), tuple(dst.shape)), hit[1])
    return dst


def add_layernorm(x: Tensor, residual: Optional[Tensor], norm: torch.nn.LayerNorm, want_rowmax: bool = True) -> Tensor:
    """LayerNorm(x + residual) over the last dimension in one launch (csrc/isg_attn.hip::add_layernorm_kernel), the
    result carrying its row maxima for the next Linear.  x / residual: fp32 [M, D] rows."""
    lib = _lib.load()
    M, D = x.shape
    if residual is not None and tuple(residual.shape) != (M, D):
        raise ValueError(f"add_layernorm: x {tuple(x.shape)} vs residual {tuple(residual.shape)}")
    if tuple(norm.normalized_shape) != (D,) or norm.weight is None:
        raise ValueError("add_layernorm: an affine LayerNorm over the last dimension")
    out = torch.empty(M, D, dtype=torch.float32, device=x.device)
    rm = torch.empty(M, 1, dtype=torch.float32, device=x.device) if want_rowmax else None
    # the consumers of a LayerNorm result are Linears: where they run on the planes32 engine the kernel writes the planes too
    pl = None
    if CFG.ln_planes and CFG.h3p and D % 32 == 0 and D >= CFG.h3p_min_k and M >= CFG.h3p_min_m:
        pl = Planes32(torch.empty(M * D * 2, dtype=torch.int16, device=x.device),
                      torch.empty(M, dtype=torch.float32, device=x.device), M, D)
    rc = lib.isg_add_layernorm(_chk_rows(x, "x"), x.stride(0), 0 if residual is None else _chk_rows(residual, "residual"),
                               0 if residual is None else residual.stride(0),
                               _chk(norm.weight.detach(), "weight", torch.float32, (D,)),
                               _chk(None if norm.bias is None else norm.bias.detach(), "bias", torch.float32, (D,), optional=True),
                               float(norm.eps), out.data_ptr(), D, 0 if rm is None else rm.data_ptr(), M, D,
                               0 if pl is None else pl.planes.data_ptr(), 0 if pl is None else pl.inv.data_ptr(), _stream())
    if rc == ISG_EUNSUPPORTED:
        COUNTERS["torch_layer_norm"] += 1
        y = x if residual is None else x + residual
        return torch.nn.functional.layer_norm(y, norm.normalized_shape, norm.weight, norm.bias, norm.eps)
    _lib.check(rc, "isg_add_layernorm")
    if rm is not None:
        attach_row_maxima(out, rm)
    if pl is not None:
        out._isg_planes32 = ((_ver(out), out.data_ptr(), tuple(out.shape)), pl)       # what split_planes32(out) would make
    return out


def _linear_torch(x: Tensor, weight: Tensor, bias: Optional[Tensor], gelu: bool, relu: bool) -> Tensor:
    COUNTERS["torch_linear"] += 1
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.relu(y) if relu else (torch.nn.functional.gelu(y) if gelu else y)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, gelu: bool = False,
           cache_planes: bool = True, out_dtype=torch.float32, relu: bool = False, want_rowmax: bool = False) -> Tensor:
    """act(x @ weight^T + bias), x [M,K] fp32, weight [N,K] (torch Linear layout).  Uses the bf16x6 matrix-core kernel
    when the shape allows it, hipBLASLt through torch otherwise (K not a multiple of 4).  ``cache_planes=False``: the
    weight is being trained (or is a temporary), so its bf16 planes are split per call instead of cached."""
    if isinstance(x, Planes32):                    # the producer handed the rows over pre-split (a Linear's planes output)
        return linear_h3p(x, weight, bias, gelu=gelu, relu=relu, cache_planes=cache_planes)
    M, K = x.shape
    if (M <= CFG.skinny_max_m and out_dtype == torch.float32 and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and skinny_supported(M, weight.size(0), K)):
        # the latency-bound regime first: at these sizes the HOST's time per call is what the forward costs, and
        # nothing below applies (inference, fp32 rows in and out)
        y = linear_skinny(x, weight, bias, gelu=gelu, relu=relu)
        if y is not None:
            return y
    f16_io = x.dtype == torch.float16 or out_dtype == torch.float16
    if f16_io and (CFG.gemm_backend != "bf16x6" or (K & 3) != 0 or _rec(x, weight, bias)):
        raise _lib.IsgError("fp16 feature rows are an inference feature of the bf16x6 kernel (K % 4 == 0, no autograd)")
    if relu and (gelu or f16_io):
        raise ValueError("relu excludes gelu and fp16 rows")
    if relu and (_rec(x, weight, bias) or CFG.gemm_backend != "bf16x6" or (K & 3) != 0 or M == 0):
        return _linear_torch(x, weight, bias, False, True)
    if _rec(x, weight, bias) and CFG.gemm_backend == "bf16x6" and (K & 3) == 0 and M > 0:
        from . import autograd
        return autograd.linear(x, weight, bias, gelu)
    N = weight.size(0)
    if CFG.gemm_backend != "bf16x6" or (K & 3) != 0 or M == 0:
        if M == 0:
            return x.new_empty(0, N)
        return _linear_torch(x, weight, bias, gelu, False)
    if skinny_supported(M, N, K) and not f16_io and x.dtype == torch.float32:
        y = linear_skinny(x, weight, bias, gelu=gelu, relu=relu)
        if y is not None:
            return y
    if (not f16_io and h3p_supported(M, N, K) and (M >= CFG.h3p_min_m_unsplit or has_planes32(x)) and x.stride(1) == 1
            and (x.stride(0) & 3) == 0 and (x.data_ptr() & 15) == 0):
        # K >= 256 over many rows: the planes32 engine (csrc/isg_gemm_h3p.hip); the split of x stays attached to x.  Between
        # CFG.h3p_min_m and CFG.h3p_min_m_unsplit rows only an input whose PRODUCER left its planes (isg_add_layernorm, the attention
        # kernels, the gates): an isolated Linear there would pay a split pass of its own and loses to the tile kernels
        return linear_h3p(x, weight, bias, gelu=gelu, relu=relu, cache_planes=cache_planes)
    lib = _lib.load()
    out = torch.empty(M, N, dtype=out_dtype, device=x.device)
    a_rowmax = row_maxima(x)
    if a_rowmax is not None and (a_rowmax.dim() != 2 or a_rowmax.size(0) != M or a_rowmax.size(1) > 64 or
                                 a_rowmax.stride(1) != 1):
        a_rowmax = None
    f16x3_tile = CFG.gemm_f16x3 and CFG.f16x3_tile and CFG.gemm_kernel == "auto" and K > 128 and not f16_io and M < (1 << 23)
    nchunk = (K + 639) // 640              # chains of at most 640: 2.75x an fp32 GEMM's error at 1024-long chains, < 2x here
    step = (K + nchunk - 1) // nchunk
    step = (step + 31) // 32 * 32
    # the producer's partial maxima serve a K-chunk when every partial covers a whole number of columns of ONE chunk
    P = a_rowmax.size(1) if a_rowmax is not None else 0
    per = K // P if P and K % P == 0 else 0
    sliced = P > 0 and (nchunk == 1 or (per > 0 and step % per == 0))      # one chunk: any partition of the row serves
    if f16x3_tile and not sliced and not (N >= 256 and M >= 4096):
        f16x3_tile = False         # a pass over x for the row maxima only pays for wide Linears over many rows
    if f16x3_tile:
        # fp16 three-product tile kernel.  Row scales: the producer's partial maxima (a chunk takes the slice that covers
        # its columns), or one pass over x which is then left on x for its other consumers; reductions longer than 640 run
        # as K-chunks that accumulate into `out` (every chunk its own fp32 chain and its own row scales)
        planes, inv = _weight_planes(weight, cache_planes, "f16x3_rows")
        act = 2 if relu else (1 if gelu else 0)
        bptr = _chk(None if bias is None else bias.detach(), "bias", torch.float32, (N,), optional=True)
        d_rowmax = torch.empty(M, (N + 31) // 32, dtype=torch.float32, device=x.device) if want_rowmax else None
        xp = _chk(x, "x", torch.float32)
        k0, c = 0, 0
        while k0 < K:
            kc = min(step, K - k0)
            last = k0 + kc >= K
            if sliced and nchunk == 1:
                rm_ptr, rm_p, rm_ld = a_rowmax.data_ptr(), P, a_rowmax.stride(0)
            elif sliced:
                rm_ptr, rm_p, rm_ld = a_rowmax.data_ptr() + 4 * (k0 // per), (kc + per - 1) // per, a_rowmax.stride(0)
            else:
                rm = torch.empty(M, 1, dtype=torch.float32, device=x.device)
                _lib.check(lib.isg_row_absmax(xp + 4 * k0, M, kc, K, rm.data_ptr(), _stream()), "isg_row_absmax")
                COUNTERS["row_absmax"] += 1
                if nchunk == 1:
                    attach_row_maxima(x, rm)          # the next Linear over the same rows does not repeat the pass
                rm_ptr, rm_p, rm_ld = rm.data_ptr(), 1, 1
            rc = lib.isg_linear_f16x3_tile(
                xp + 4 * k0, rm_ptr, rm_p, rm_ld, planes.data_ptr(), inv.data_ptr(), bptr if last else 0,
                out.data_ptr(), d_rowmax.data_ptr() if (last and d_rowmax is not None) else 0, M, N, kc, K, N,
                act if last else 0, K, k0, 1 if c > 0 else 0, _stream())
            if rc == ISG_EUNSUPPORTED and c == 0:       # a shape the kernel has no launch for (> 65535 row tiles): hipBLASLt
                return _linear_torch(x, weight, bias, gelu, relu)
            _lib.check(rc, "isg_linear_f16x3_tile")
            k0 += kc
            c += 1
        if d_rowmax is not None:
            attach_row_maxima(out, d_rowmax)
        return out
    if (_use_panel(M, N, K) and not relu and CFG.gemm_f16x3 and K <= 128 and CFG.f16x3_f16_out and x.dtype == torch.float32
            and out_dtype == torch.float16):
        # fp32 rows in, half rows out (configs[4]'s x_l | x_r): the three-product kernel with one rounding at its store
        planes, inv = _weight_planes(weight, cache_planes, "f16x3")
        _lib.check(lib.isg_linear_f16x3_f16(
            _chk(x, "x", torch.float32), planes.data_ptr(), inv.data_ptr(),
            _chk(None if bias is None else bias.detach(), "bias", torch.float32, (N,), optional=True),
            out.data_ptr(), M, N, K, K, N, 1 if gelu else 0, N, 0, _stream()), "isg_linear_f16x3_f16")
        return out
    if _use_panel(M, N, K) and not relu and CFG.gemm_f16x3 and K <= 128 and not f16_io:
        planes, inv = _weight_planes(weight, cache_planes, "f16x3")
        _lib.check(lib.isg_linear_f16x3(
            _chk(x, "x", torch.float32), planes.data_ptr(), inv.data_ptr(),
            _chk(None if bias is None else bias.detach(), "bias", torch.float32, (N,), optional=True),
            out.data_ptr(), M, N, K, K, N, 1 if gelu else 0, N, 0, _stream()), "isg_linear_f16x3")
        return out
    if _use_panel(M, N, K) and not relu:
        planes = _weight_planes(weight, cache_planes, "panel")
        _lib.check(lib.isg_linear_panel(
            _chk(x, "x", x.dtype), 1 if x.dtype == torch.float16 else 0, planes.data_ptr(),
            _chk(None if bias is None else bias.detach(), "bias", torch.float32, (N,), optional=True),
            out.data_ptr(), 1 if out_dtype == torch.float16 else 0, M, N, K, K, N, 1 if gelu else 0, _stream()),
            "isg_linear_panel")
        return out
    planes = _weight_planes(weight, cache_planes)
    if f16_io:
        if M > 0:
            _lib.check(lib.isg_linear_bf16x6_f16(
                _chk(x, "x", x.dtype), 1 if x.dtype == torch.float16 else 0, planes.data_ptr(),
                _chk(None if bias is None else bias.detach(), "bias", torch.float32, (N,), optional=True),
                out.data_ptr(), 1 if out_dtype == torch.float16 else 0, M, N, K, K, N, 1 if gelu else 0, _stream()),
                "isg_linear_bf16x6_f16")
        return out
    rc = lib.isg_linear_bf16x6(_chk(x, "x", torch.float32), planes.data_ptr(),
                               _chk(None if bias is None else bias.detach(), "bias", torch.float32, (N,), optional=True),
                               out.data_ptr(), M, N, K, K, N, 2 if relu else (1 if gelu else 0), _stream())
    if rc == ISG_EUNSUPPORTED:                          # e.g. more than 65535 row tiles: fp32 through hipBLASLt
        return _linear_torch(x, weight, bias, gelu, relu)
    _lib.check(rc, "isg_linear_bf16x6")
    return out


# ------------------------------------------------------------------------------------------------
# The K >= 256 engine (csrc/isg_gemm_h3p.hip): both operands pre-split ("planes32"), LDS-DMA operand path
# ------------------------------------------------------------------------------------------------
class Planes32(NamedTuple):
    """An activation as the fp16 three-product GEMM reads it: planes [rows * ceil(cols / 32) * 64] int16 (row r, k-tile kt =
    one 128-byte line [hi 32 | mid 32] of the row scaled by its own power of two), inv [rows] = 1 / scale."""
    planes: Tensor
    inv: Tensor
    rows: int
    cols: int
    # SEGMENTED (isg_gatv2_mp_fwd_planes): columns [0, seg_cols) of a row under inv_first, the rest under inv; each segment padded
    # to whole 32-column lines, so the planes hold 2 * ceil(seg_cols / 32) lines per row and the weight is laid out to match
    inv_first: Optional[Tensor] = None
    seg_cols: int = 0


def has_planes32(x: Tensor) -> bool:
    """Did x's producer leave its planes32 (still valid for x's current version)?"""
    hit = getattr(x, "_isg_planes32", None)
    return hit is not None and hit[0] == (_ver(x), x.data_ptr(), tuple(x.shape))


def split_planes32(x: Tensor) -> Planes32:
    """fp32 rows [M, K] -> planes32 (exact row maxima).  The planes stay attached to `x` (tied to its version counter, like
    the row maxima): a second Linear over the same rows (the decoder layers' cross-attention over the encoder memory) does
    not repeat the pass."""
    hit = getattr(x, "_isg_planes32", None)
    if hit is not None and hit[0] == (_ver(x), x.data_ptr(), tuple(x.shape)):
        return hit[1]
    lib = _lib.load()
    M, K = x.shape
    planes = torch.empty(int(lib.isg_planes32_elems(M, K)), dtype=torch.int16, device=x.device)
    inv = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.check(lib.isg_split_planes32(_chk_rows(x, "x"), M, K, x.stride(0), planes.data_ptr(), inv.data_ptr(), _stream()),
               "isg_split_planes32")
    out = Planes32(planes, inv, M, K)
    x._isg_planes32 = ((_ver(x), x.data_ptr(), tuple(x.shape)), out)
    return out


def instr_gate_planes32(x: Tensor, instr: Tensor, batch: Tensor, want_rows: bool = False):
    """gelu(x * instr[batch]) (mgat_v2_conv.py:156-157) as Planes32 for the lin_l | lin_r projection on the engine, plus the fp32
    rows when a masked layer's node gate needs them (they then carry the planes: a Linear over them splits nothing).
    Returns (rows or None, Planes32).  Inference only."""
    lib = _lib.load()
    N, C = x.shape
    rows = torch.empty_like(x) if want_rows else None
    planes = torch.empty(max(int(lib.isg_planes32_elems(N, C)), 1), dtype=torch.int16, device=x.device)
    inv = torch.empty(max(N, 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.isg_instr_gate_planes32(_chk(x, "x", torch.float32), _chk(instr, "instr", torch.float32, (instr.size(0), C)),
                                           _chk(batch, "batch", torch.int64, (N,)), 0 if rows is None else rows.data_ptr(),
                                           planes.data_ptr(), inv.data_ptr(), N, C, _stream()), "isg_instr_gate_planes32")
    pl = Planes32(planes, inv, N, C)
    if rows is not None:
        rows._isg_planes32 = ((_ver(rows), rows.data_ptr(), tuple(rows.shape)), pl)
    return rows, pl


def _h3p_weight(weight: Tensor, bias: Optional[Tensor], cache: bool = True, seg_cols: int = 0):
    """planes32 of a weight [N, K], its inverse row scales and the output bound {2^14 * max_n ||w_n||_1, max |b|} (device).
    seg_cols: the layout of a SEGMENTED activation -- columns [0, seg_cols) and [seg_cols, K) each padded with zero columns to a
    multiple of 32."""
    def build():
        w = weight.detach()
        if seg_cols:
            sp = (seg_cols + 31) // 32 * 32
            wl = torch.zeros(w.size(0), 2 * sp, dtype=w.dtype, device=w.device)
            wl[:, :seg_cols] = w[:, :seg_cols]
            wl[:, sp:sp + w.size(1) - seg_cols] = w[:, seg_cols:]
            p = split_planes32(wl)
        else:
            p = split_planes32(w.contiguous().clone())       # a private copy: nothing stays attached to the parameter
        l1 = w.abs().sum(dim=1).max() * 16384.0
        bm = bias.detach().abs().max() if bias is not None else torch.zeros((), device=w.device)
        return p.planes, p.inv, torch.stack([l1, bm]).to(torch.float32).contiguous()
    if not cache:
        with torch.no_grad():
            return build()
    return derived_weight(f"h3p{seg_cols or ''}", (weight,) if bias is None else (weight, bias), build)


# CFG.rows_kernel_min_edges (ops.ROWS_KERNEL_MIN_EDGES): below this many edges a wide layer (C = 300 / K = 300) projects its edge rows and runs un-fused
# CFG.skinny (ops.SKINNY): Linears over at most CFG.skinny_max_m rows (and M N K <= CFG.skinny_max_work) on isg_linear_skinny (A/B switch)
# CFG.skinny_max_m (ops.SKINNY_MAX_M): the latency-bound regime: a handful of questions per forward (csrc/isg_gemm_skinny.hip)
# CFG.skinny_max_work (ops.SKINNY_MAX_WORK): true fp32 MFMAs run at 1/16 of the fp16 rate: beyond ~1e9 multiply-adds the exact-split tile kernels win
def skinny_supported(M: int, N: int, K: int) -> bool:
    return (CFG.skinny and CFG.gemm_backend == "bf16x6" and CFG.gemm_kernel == "auto" and 0 < M <= CFG.skinny_max_m and (K & 3) == 0
            and M * N * K <= CFG.skinny_max_work)


def linear_skinny(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, gelu: bool = False, relu: bool = False) -> Optional[Tensor]:
    """act(x @ weight^T + bias) on isg_linear_skinny (small M: the reduction split over a workgroup's waves, true fp32 MFMAs, the
    weight read as it is).  None when the operands' layout is not the kernel's (the caller takes the tile kernels)."""
    # (host time matters here -- a forward at this size is ~70 of these calls and the GPU needs ~10 us for each: every tensor
    # property is read once, nothing is detached or viewed)
    if x.dim() != 2 or weight.dim() != 2:
        return None
    M, K = x.shape
    N, Kw = weight.shape
    lda, ldw, xp, wp = x.stride(0), weight.stride(0), x.data_ptr(), weight.data_ptr()
    if (x.stride(1) != 1 or weight.stride(1) != 1 or ((lda | ldw) & 3) or ((xp | wp) & 15) or x.dtype != torch.float32
            or weight.dtype != torch.float32 or weight.device != x.device):
        return None
    if not x.is_cuda:
        raise _lib.IsgError(f"x must live on the GPU (got {x.device}); this path has no CPU fallback")
    if Kw != K:
        raise ValueError(f"linear_skinny: x has {K} columns, weight {tuple(weight.shape)}")
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        raise NotImplementedError("linear_skinny has no backward (autograd.linear is the differentiable Linear)")
    bp = 0
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous() or bias.device != x.device:
            raise ValueError(f"linear_skinny: bias must be a contiguous fp32 [{N}] tensor on {x.device}")
        bp = bias.data_ptr()
    lib = _lib.load()
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    rc = lib.isg_linear_skinny(xp, lda, wp, ldw, bp, out.data_ptr(), N, M, N, K, 2 if relu else (1 if gelu else 0), _stream())
    if rc == ISG_EUNSUPPORTED:
        return None
    _lib.check(rc, "isg_linear_skinny")
    COUNTERS["linear_skinny"] += 1
    return out


# CFG.h3p (ops.H3P): Linears with K >= CFG.h3p_min_k over at least CFG.h3p_min_m rows on isg_linear_h3p (A/B switch)
# CFG.h3p_min_k (ops.H3P_MIN_K): CFG.ln_planes (ops.LN_PLANES): isg_add_layernorm writes its result as planes32 too where Linears on the engine read it (A/B switch)
# CFG.h3p_chain (ops.H3P_CHAIN): linear1 -> linear2 of the Transformer layers through planes (no fp32 intermediate): A/B switch
# CFG.h3p_min_m_unsplit (ops.H3P_MIN_M_UNSPLIT): ... but a Linear whose input carries no planes from its producer takes the engine from this many rows only
# CFG.h3p_min_m (ops.H3P_MIN_M): 8192 until round 6; swept with the small-batch kernels in place (tools/time_full_model.py G --set=H3P_MIN_M=...): 2048 is 5-7 % faster at 160-700 graphs
def h3p_supported(M: int, N: int, K: int) -> bool:
    return (CFG.h3p and CFG.gemm_backend == "bf16x6" and CFG.gemm_kernel == "auto" and CFG.gemm_f16x3 and K >= CFG.h3p_min_k and (K & 3) == 0 and
            (N & 3) == 0 and M >= CFG.h3p_min_m and M * ((K + 31) // 32) * 128 < (1 << 31) and N * ((K + 31) // 32) * 128 < (1 << 31)
            and M * ((N + 31) // 32 * 32) * 4 < (1 << 32) - 16)       # the result through a buffer descriptor: 32-bit byte offsets


# The cache policy of a large (>= 128 MB) fp32 result's stores in isg_linear_h3p: -1 (the library's choice: nt at K >= 512, plain
# below) / 0 plain / 1 nt / 2 sc0 sc1 nt, or "auto" = measured once per process on the box (_h3p_tune).  Results do not depend on
# it.  In ISOLATION 2 runs the K = 300 projections 15-32 % faster than plain stores on some MI355X boxes and 21 % slower on others
# (the same way in every process on a box); in the full model, on a box where it wins in isolation, every policy gives the same
# step (20.09-20.29 ms: profiles/r04_ag_h3p_store_policy.txt) -- what the producer gains by not leaving its result in L2 / the
# Infinity Cache its consumer loses.  So the default is the library's choice and "auto" stays an experiment.
# CFG.h3p_store_policy (ops.H3P_STORE_POLICY): 
# CFG.linear_multi_h3p (ops.LINEAR_MULTI_H3P): the layers' lin_edge over the shared edge features as one engine launch (A/B switch)
# CFG.mp_planes (ops.MP_PLANES): the flat message-passing kernel hands x_proj.0 its operand as segmented planes32 (A/B switch)
# CFG.gather_add_planes (ops.GATHER_ADD_PLANES): isg_gather_add hands its rows to the Linear behind it as planes32 (A/B switch)
_h3p_policy_state = {"chosen": None, "us": None}


def h3p_store_policy():
    """What _h3p_tune chose in this process (None before the first large Linear), with the times it measured."""
    return dict(_h3p_policy_state)


def _h3p_tune(dev) -> None:
    """One-time choice of the store policy for large results: the K = 300 projection of 65 536 rows onto 1 200 columns (a 315 MB
    result), the library's own choice against write-through streaming stores, median of three launches each.  ~3 ms, once per
    process; skipped (library's choice) while a stream is being captured."""
    lib = _lib.load()
    if CFG.h3p_store_policy != "auto":
        _lib.check(lib.isg_linear_h3p_store_policy(int(CFG.h3p_store_policy)), "isg_linear_h3p_store_policy")
        _h3p_policy_state.update(chosen=int(CFG.h3p_store_policy), us=None)
        return
    if torch.cuda.is_current_stream_capturing():
        return
    M, N, K = 65536, 1200, 300
    KT = (K + 31) // 32
    ap = torch.zeros(M * KT * 64, dtype=torch.int16, device=dev)
    ainv = torch.ones(M, dtype=torch.float32, device=dev)
    wp = torch.zeros(N * KT * 64, dtype=torch.int16, device=dev)
    winv = torch.ones(N, dtype=torch.float32, device=dev)
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ap.random_(0, 15000)          # fp16 bit patterns of finite values in [0, 0.8): the kernel's speed does not depend on the
    wp.random_(0, 15000)          # values, but the chip's clocks do on all-zero operands
    us = {}
    for pol in (-1, 2, -1, 2):
        _lib.check(lib.isg_linear_h3p_store_policy(pol), "isg_linear_h3p_store_policy")
        ts = []
        for _ in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            _lib.check(lib.isg_linear_h3p(ap.data_ptr(), ainv.data_ptr(), wp.data_ptr(), winv.data_ptr(), 0, out.data_ptr(), 0, 0, 0,
                                          M, N, K, N, 0, 0, 0, _stream()), "isg_linear_h3p")
            e.record()
            e.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        us.setdefault(pol, []).extend(ts[1:])
    med = {k: sorted(v)[len(v) // 2] for k, v in us.items()}
    chosen = 2 if med[2] < 0.95 * med[-1] else -1
    _lib.check(lib.isg_linear_h3p_store_policy(chosen), "isg_linear_h3p_store_policy")
    _h3p_policy_state.update(chosen=chosen, us={str(k): round(v, 1) for k, v in med.items()})


def linear_h3p(x, weight: Tensor, bias: Optional[Tensor] = None, gelu: bool = False, relu: bool = False,
               planes_out: bool = False, cache_planes: bool = True):
    """act(x @ weight^T + bias) on the planes32 engine.  x: Planes32 or fp32 rows (split here, once per tensor version).
    planes_out: the result as Planes32 (its columns padded with zeros to a multiple of 32), scaled by the bound known before the
    product -- the input of the next Linear with no pass over it; otherwise fp32 [M, N]."""
    lib = _lib.load()
    xp = x if isinstance(x, Planes32) else split_planes32(x)
    M, K = xp.rows, xp.cols
    N = weight.size(0)
    if weight.size(1) != K:
        raise ValueError(f"linear_h3p: x has {K} columns, weight {tuple(weight.shape)}")
    seg = xp.seg_cols
    if seg and (not gelu or relu or 2 * seg != K):
        raise ValueError("linear_h3p: a segmented operand (isg_gatv2_mp_fwd_planes) feeds a Linear + GELU over two equal halves")
    wp, winv, bound = _h3p_weight(weight, bias, cache_planes, seg)
    bptr = _chk(None if bias is None else bias.detach(), "bias", torch.float32, (N,), optional=True)
    act = 2 if relu else (1 if gelu else 0)
    dev = xp.planes.device
    ksplit = (seg + 31) // 32 * 32
    Kc = 2 * ksplit if seg else K                    # the k extent the kernel walks: both segments with their padding
    seg_args = (xp.inv_first.data_ptr(), ksplit) if seg else (0, 0)
    timer = H3P_TIMER
    COUNTERS["linear_h3p"] += 1                      # launches of the engine / of them, those fed a segmented message-passing result
    COUNTERS["h3p_segmented"] += 1 if seg else 0
    if timer is not None:
        ev0, ev1 = timer.bracket({"M": M, "N": N, "K": Kc, "planes_out": bool(planes_out)})
    if planes_out:
        dp = torch.empty(int(lib.isg_planes32_elems(M, N)), dtype=torch.int16, device=dev)
        dinv = torch.empty(M, dtype=torch.float32, device=dev)
        if timer is not None:
            ev0.record()
        _lib.check(lib.isg_linear_h3p(xp.planes.data_ptr(), xp.inv.data_ptr(), wp.data_ptr(), winv.data_ptr(), bptr, 0,
                                      dp.data_ptr(), dinv.data_ptr(), bound.data_ptr(), M, N, Kc, 0, act, *seg_args, _stream()),
                   "isg_linear_h3p")
        if timer is not None:
            ev1.record()
        return Planes32(dp, dinv, M, N)
    if _h3p_policy_state["chosen"] is None and M * N * 4 >= 128_000_000:
        _h3p_tune(dev)
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    if timer is not None:
        ev0.record()
    _lib.check(lib.isg_linear_h3p(xp.planes.data_ptr(), xp.inv.data_ptr(), wp.data_ptr(), winv.data_ptr(), bptr,
                                  out.data_ptr(), 0, 0, 0, M, N, Kc, N, act, *seg_args, _stream()), "isg_linear_h3p")
    if timer is not None:
        ev1.record()
    return out


def planes32_to_rows(p: Planes32) -> Tensor:
    """fp32 rows of a Planes32 (hi + mid) * inv: tests and diagnostics only."""
    if p.seg_cols:
        st = (p.seg_cols + 31) // 32
        v = p.planes.view(torch.float16).view(p.rows, 2 * st, 2, 32).float()
        full = (v[:, :, 0] + v[:, :, 1]).reshape(p.rows, 2, st * 32)
        return torch.cat([full[:, 0, :p.seg_cols] * p.inv_first[:, None], full[:, 1, :p.cols - p.seg_cols] * p.inv[:, None]],
                         dim=1).contiguous()
    KT = (p.cols + 31) // 32
    v = p.planes.view(torch.float16).view(p.rows, KT, 2, 32).float()
    return ((v[:, :, 0] + v[:, :, 1]).reshape(p.rows, KT * 32)[:, :p.cols] * p.inv[:, None]).contiguous()


# CFG.linear_multi (ops.LINEAR_MULTI): A/B switch (tools/ab_step.py)


def linear_multi(x: Tensor, weights, out_dtype=torch.float32):
    """x @ W_i^T for several bias-free Linears of one shape over the same rows, as ONE launch of the row-panel kernel
    (isg_linear_panel_multi): a tuple of dense [M, n] tensors, or None when the shape is not the panel kernel's (the caller
    then projects layer by layer).  MGAT's per-layer lin_edge projections of the shared edge features use it."""
    if not CFG.linear_multi or CFG.gemm_backend != "bf16x6" or _rec(x, *weights) or len(weights) < 2:
        return None
    M, K = x.shape
    n = weights[0].size(0)
    if any(tuple(w.shape) != (n, K) for w in weights):
        return None
    if (CFG.linear_multi_h3p and x.dtype == torch.float32 and out_dtype == torch.float32 and (n & 3) == 0 and
            h3p_supported(M, len(weights) * n, K) and x.stride(1) == 1 and (x.stride(0) & 3) == 0 and (x.data_ptr() & 15) == 0):
        # K >= 256 (the reference's default width): ONE launch of the planes32 engine over the concatenated weights -- the shared
        # rows are read once instead of once per layer (262 MB per layer at 205 k edges); the layers' results are column slices
        cat = derived_weight("linear_multi", tuple(weights), lambda: torch.cat([w.detach() for w in weights], 0).contiguous())
        y = linear_h3p(x, cat, None)
        return tuple(y[:, i * n:(i + 1) * n] for i in range(len(weights)))
    if (n & 31) or (K & 3) or not _use_panel(M, len(weights) * n, K):
        return None
    lib = _lib.load()
    cat = derived_weight("linear_multi", tuple(weights), lambda: torch.cat([w.detach() for w in weights], 0).contiguous())
    L = len(weights)
    out = torch.empty(L, M, n, dtype=out_dtype, device=x.device)
    if CFG.gemm_f16x3 and K <= 128 and x.dtype == torch.float32 and (out_dtype == torch.float32 or
                                                                 (out_dtype == torch.float16 and CFG.f16x3_f16_out)):
        planes, inv = _weight_planes(cat, True, "f16x3")
        fn = lib.isg_linear_f16x3 if out_dtype == torch.float32 else lib.isg_linear_f16x3_f16
        _lib.check(fn(_chk(x, "x", torch.float32), planes.data_ptr(), inv.data_ptr(), 0, out.data_ptr(),
                      M, L * n, K, K, n, 0, n, M * n, _stream()), "isg_linear_f16x3")
        return tuple(out[i] for i in range(L))
    planes = _weight_planes(cat, True, "panel")
    _lib.check(lib.isg_linear_panel_multi(
        _chk(x, "x", x.dtype), 1 if x.dtype == torch.float16 else 0, planes.data_ptr(), 0, out.data_ptr(),
        1 if out_dtype == torch.float16 else 0, M, L * n, K, K, n, 0, n, M * n, _stream()), "isg_linear_panel_multi")
    return tuple(out[i] for i in range(L))


def mha_small_supported(t_kv: int, head_dim: int) -> bool:
    """isg_mha_small's launch limits (csrc/isg_attn.hip): head_dim <= 64 and a multiple of 4, <= 128 keys, and a head's Q / K /
    V rows (as many queries as keys at most: the callers pass the longer of the two) + score strips within 64 KB of LDS -- at
    head_dim 64 that is 80 keys (CLIP questions: 77).  Callers ask BEFORE choosing the kernel path."""
    return head_dim <= 64 and head_dim % 4 == 0 and t_kv <= 128 and (t_kv * (3 * head_dim + 4) + 4 * 128) * 4 <= 64 * 1024


# CFG.tile_heavy_first (ops.TILE_HEAVY_FIRST): persistent tile kernels walk the tile list heavy tiles first (A/B switch)
# CFG.mha_rows_planes (ops.MHA_ROWS_PLANES): attention results as planes32 where the all-heads form fits (A/B switch)
# CFG.mha_rows_max_tq (ops.MHA_ROWS_MAX_TQ): ... up to this many query rows per batch item


def mha_rows_supported(t_q: int, t_kv: int, heads: int, head_dim: int) -> bool:
    """The all-heads form of isg_mha_small (one workgroup per batch item, the result rows assembled in LDS and written as
    planes32): a head's Q / K / V, the score strips and t_q whole rows within 64 KB -- 12-token questions and the decoder's 4
    queries at d = 512, not CLIP's 77 tokens."""
    return (CFG.mha_rows_planes and t_q <= CFG.mha_rows_max_tq and mha_small_supported(max(t_q, t_kv), head_dim) and
            (t_kv * (2 * head_dim + 4) + t_q * head_dim + (4 if t_q <= 4 else 8 if t_q <= 8 else 12) * 128 +
             t_q * heads * head_dim) * 4 <= 64 * 1024)


def mha_small(q: Tensor, k: Tensor, v: Tensor, batch_size: int, heads: int, key_bias: Optional[Tensor] = None,
              want_rowmax: bool = False, planes_out: bool = False):
    """softmax(Q K^T / sqrt(hd) + key_bias) V per (batch item, head) for short sequences (csrc/isg_attn.hip).
    q [Tq*B, D], k / v [Tk*B, D] in torch's [T, B, D] row order (row t*B + b; column slices of a fused projection are
    fine), key_bias fp32 [B, Tk] additive (question_encoder.py:35-37) -> [Tq*B, D].
    planes_out (callers ask mha_rows_supported first): the result as Planes32 only -- out_proj's operand, no split pass."""
    lib = _lib.load()
    B, H, D = int(batch_size), int(heads), q.size(1)
    hd = D // H
    Tq, Tk = q.size(0) // B, k.size(0) // B
    if H * hd != D or Tq * B != q.size(0) or Tk * B != k.size(0) or tuple(v.shape) != tuple(k.shape):
        raise ValueError(f"mha_small: q {tuple(q.shape)}, k {tuple(k.shape)}, v {tuple(v.shape)} vs B={B}, H={H}")
    kb = _chk(key_bias, "key_bias", torch.float32, (B, Tk), optional=True)
    if planes_out:
        pl = torch.empty(int(lib.isg_planes32_elems(Tq * B, D)), dtype=torch.int16, device=q.device)
        pinv = torch.empty(Tq * B, dtype=torch.float32, device=q.device)
        _lib.check(lib.isg_mha_small(_chk_rows(q, "q"), q.stride(0), _chk_rows(k, "k"), k.stride(0), _chk_rows(v, "v"),
                                     v.stride(0), kb, 0, D, 0, B, H, hd, Tq, Tk, pl.data_ptr(), pinv.data_ptr(), _stream()),
                   "isg_mha_small")
        return Planes32(pl, pinv, Tq * B, D)
    out = torch.empty(Tq * B, D, dtype=torch.float32, device=q.device)
    rm = torch.empty(Tq * B, H, dtype=torch.float32, device=q.device) if want_rowmax else None
    _lib.check(lib.isg_mha_small(_chk_rows(q, "q"), q.stride(0), _chk_rows(k, "k"), k.stride(0), _chk_rows(v, "v"),
                                 v.stride(0), kb, out.data_ptr(), D, 0 if rm is None else rm.data_ptr(), B, H, hd, Tq, Tk, 0, 0,
                                 _stream()), "isg_mha_small")
    if rm is not None:
        attach_row_maxima(out, rm)
    return out


def linear_wgrad(grad_out: Tensor, x: Tensor) -> Tensor:
    """dW[N,K] = grad_out^T x for y = x W^T: split-M GEMM on the fp32 matrix cores (csrc/isg_wgrad.hip)."""
    lib = _lib.load()
    M, N = grad_out.shape
    K = x.size(1)
    if M == 0:
        return torch.zeros(N, K, dtype=torch.float32, device=x.device)
    splits = int(lib.isg_linear_wgrad_splits(M, N, K))
    part = torch.empty(splits, N, K, dtype=torch.float32, device=x.device)
    _lib.check(lib.isg_linear_wgrad(_chk(grad_out, "grad_out", torch.float32, (M, N)), _chk(x, "x", torch.float32, (M, K)),
                                    part.data_ptr(), M, N, K, N, K, splits, _stream()), "isg_linear_wgrad")
    return part.sum(0) if splits > 1 else part[0]


def cat_mul(a: Tensor, b: Tensor) -> Tensor:
    """cat((a, b, a * b), dim=1) (isubgvqa.py:288-291).  Inference on fp32 rows: one launch that also leaves the result's row
    maxima on it for the Linear that follows (isg_cat_mul_rowmax); otherwise the torch ops."""
    if (_rec(a, b) or a.dtype != torch.float32 or b.dtype != torch.float32 or a.dim() != 2 or a.shape != b.shape
            or a.size(1) % 4 != 0 or not a.is_cuda):
        return torch.cat((a, b, a * b), dim=1)
    lib = _lib.load()
    M, C = a.shape
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty(M, 3 * C, dtype=torch.float32, device=a.device)
    rm = torch.empty(M, 1, dtype=torch.float32, device=a.device)
    rc = lib.isg_cat_mul_rowmax(a.data_ptr(), b.data_ptr(), out.data_ptr(), rm.data_ptr(), M, C, _stream())
    if rc == ISG_EUNSUPPORTED:
        return torch.cat((a, b, a * b), dim=1)
    _lib.check(rc, "isg_cat_mul_rowmax")
    return attach_row_maxima(out, rm)


def mlp(seq: torch.nn.Sequential, x: Tensor, want_rowmax: bool = False) -> Tensor:   # x may be fp16 feature rows; the output is fp32
    """Run an nn.Sequential of Linear / GELU / Dropout(eval) modules with every Linear(+GELU) pair as one launch.
    ``want_rowmax``: the caller feeds the result to another Linear (the last launch's epilogue leaves its row maxima on it)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear) or (hasattr(m, "weight") and hasattr(m, "bias") and m.weight.dim() == 2):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.GELU) and mods[i + 1].approximate == "none"
            nxt = i + (2 if fuse else 1)
            more = nxt < len(mods) and hasattr(mods[nxt], "weight") and getattr(mods[nxt].weight, "dim", lambda: 0)() == 2
            if not isinstance(x, Planes32):
                rm = row_maxima(x)
                x = x.contiguous()
                if rm is not None:
                    attach_row_maxima(x, rm)
            tail = want_rowmax and not any(hasattr(t, "weight") and getattr(t.weight, "dim", lambda: 0)() == 2 for t in mods[nxt:]) \
                and all(isinstance(t, torch.nn.Dropout) and not t.training for t in mods[nxt:])   # only identities follow
            rows_in = x.rows if isinstance(x, Planes32) else x.size(0)
            chain = (more and CFG.h3p_chain and not _rec(m.weight, mods[nxt].weight) and not torch.is_grad_enabled() and
                     (isinstance(x, Planes32) or x.dtype == torch.float32) and
                     h3p_supported(rows_in, m.weight.size(0), m.weight.size(1)) and
                     h3p_supported(rows_in, mods[nxt].weight.size(0), mods[nxt].weight.size(1)))
            if chain:      # this Linear's result as the planes the next Linear reads: no fp32 intermediate, no split pass
                x = linear_h3p(x, m.weight, m.bias, gelu=fuse, planes_out=True)
            else:
                x = linear(x, m.weight, m.bias, gelu=fuse, want_rowmax=more or tail)      # the epilogue's maxima cost next to nothing
            i = nxt
        else:
            x = m(x)
            i += 1
    return x


_CAT = {}   # (ids of the weights) -> (versions, concatenated weight, concatenated bias, weakrefs of the weights)


def linear_fused(x: Tensor, layers, out_dtype=torch.float32) -> Tuple[Tensor, ...]:
    """Several Linear layers that share their input as ONE projection (weights concatenated along the output dim, cached);
    returns one column-slice view of the fused output per layer (row stride = total width)."""
    if _rec(*([] if isinstance(x, Planes32) else [x]), *[m.weight for m in layers]):      # training: differentiable concatenation, nothing cached
        w = torch.cat([m.weight for m in layers], dim=0)
        b = torch.cat([m.bias if m.bias is not None else torch.zeros(m.weight.size(0), device=w.device)
                       for m in layers]) if any(m.bias is not None for m in layers) else None
        y = linear(x, w, b)
        outs, o = [], 0
        for m in layers:
            n = m.weight.size(0)
            outs.append(y[:, o:o + n])
            o += n
        return tuple(outs)
    key = tuple(id(m.weight) for m in layers)
    ver = tuple((_ver(m.weight), m.weight.data_ptr(), None if m.bias is None else _ver(m.bias)) for m in layers)
    hit = _CAT.get(key)
    if hit is None or hit[0] != ver or any(r() is not m.weight for r, m in zip(hit[3], layers)):
        w = torch.cat([m.weight.detach() for m in layers], dim=0).contiguous()
        has_b = any(m.bias is not None for m in layers)
        b = torch.cat([m.bias.detach() if m.bias is not None else torch.zeros(m.weight.size(0), device=w.device)
                       for m in layers]) if has_b else None
        if len(_CAT) > 256:
            for k in [k for k, v in _CAT.items() if any(r() is None for r in v[3])]:
                del _CAT[k]
        hit = (ver, w, b, tuple(weakref.ref(m.weight) for m in layers), _Ready(w.is_cuda))
        _CAT[key] = hit
    else:
        hit[4].wait()
    y = linear(x, hit[1], hit[2], out_dtype=out_dtype)
    outs, o = [], 0
    for m in layers:
        n = m.weight.size(0)
        outs.append(y[:, o:o + n])
        o += n
    return tuple(outs)
