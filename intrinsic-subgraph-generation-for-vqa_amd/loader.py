"""ctypes binding of libisg_loader.so (include/isg_loader.h) and the batch object it fills.

Host side of SURVEY §8f row 2: GQA scene-graph JSON -> token tensors (once per image, while parsing) -> collated PyG-style
batch written straight into pinned host memory -> one async copy per tensor -> `ops.GraphPlan` without a device sync
(the collate also returns the max nodes / edges per graph).  Reference: ISubGVQA/datasets/scene_graph.py:145-389,
ISubGVQA/datasets/gqa.py:170-175,237-272.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_int, c_int64, c_void_p
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

ABI_VERSION = 1
_HERE = os.path.dirname(os.path.abspath(__file__))
# ISG_LOADER_LIB: an alternative build of the same library (the sanitizer build of tools/asan_loader.sh)
LIB_PATH = os.environ.get("ISG_LOADER_LIB") or os.path.join(_HERE, "csrc", "libisg_loader.so")

SIGNATURES = {
    "isg_loader_abi_version": (c_int, []),
    "isg_sg_last_error": (c_char_p, []),
    "isg_sg_vocab_build": (c_int, [c_void_p, c_int64, c_void_p]),
    "isg_sg_vocab_size": (c_int64, [c_void_p]),
    "isg_sg_vocab_lookup": (c_int64, [c_void_p, c_char_p]),
    "isg_sg_vocab_free": (None, [c_void_p]),
    "isg_sg_store_create": (c_int, [c_void_p, c_void_p]),
    "isg_sg_store_add_json": (c_int, [c_void_p, c_char_p, c_int64]),
    "isg_sg_store_add_json_file": (c_int, [c_void_p, c_char_p]),
    "isg_sg_store_num_graphs": (c_int64, [c_void_p]),
    "isg_sg_store_find": (c_int64, [c_void_p, c_char_p]),
    "isg_sg_store_free": (None, [c_void_p]),
    "isg_sg_store_find_many": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "isg_sg_collate_sizes": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "isg_sg_collate": (c_int, [c_void_p, c_void_p, c_int64] + [c_void_p] * 8 + [ctypes.c_int32]),
}

_lib = None


class LoaderError(RuntimeError):
    pass


def load():
    """Load libisg_loader.so (built by __graft_entry__.build()); fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LoaderError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.isg_loader_abi_version() != ABI_VERSION:
        raise LoaderError(f"libisg_loader.so ABI {lib.isg_loader_abi_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise LoaderError(f"{what} failed ({rc}): {load().isg_sg_last_error().decode('utf-8', 'replace')}")


def _c_strings(items: Sequence[str]):
    arr = (c_char_p * len(items))(*[s.encode("utf-8") for s in items])
    return arr


class SceneGraphVocab:
    """The scene-graph token vocabulary (scene_graph.py:145-183): ``token_lists`` are the six lists the reference reads
    from meta_info (name, attr, rel text files; objects, predicates, attributes JSON), in that order."""

    def __init__(self, token_lists: Sequence[Sequence[str]]):
        lib = load()
        flat = [t for lst in token_lists for t in lst]
        self._h = c_void_p()
        arr = _c_strings(flat)
        _check(lib.isg_sg_vocab_build(arr, len(flat), ctypes.byref(self._h)), "isg_sg_vocab_build")
        self._tokens = list(dict.fromkeys(flat + ["<self>", "pokemon"]))

    def __len__(self) -> int:
        return int(load().isg_sg_vocab_size(self._h))

    def lookup(self, token: str) -> int:
        return int(load().isg_sg_vocab_lookup(self._h, token.encode("utf-8")))

    def get_stoi(self) -> Dict[str, int]:
        out = {t: self.lookup(t) for t in ["<unk>", "<pad>", "<sos>", "<eos>", "<self>"] + self._tokens}
        return {t: i for t, i in out.items() if i >= 0}

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.isg_sg_vocab_free(self._h)
            self._h = None


@dataclass
class SceneGraphBatch:
    """What torch_geometric's Batch.from_data_list hands to ISubGVQA.forward (gqa.py:258), plus the
    per-graph bounds that let ops.GraphPlan.build skip its device sync."""
    x: Tensor               # [N, 4] int64 name + 3 attribute tokens
    edge_index: Tensor      # [2, E] int64
    edge_attr: Tensor       # [E] int64
    x_bbox: Tensor          # [N, 4] int64
    added_sym_edge: Tensor  # [S] int64, per-graph edge positions (not offset, quirk Q6)
    batch: Tensor           # [N] int64
    ptr: Tensor             # [B+1] int64
    num_graphs: int
    max_nodes: int
    max_edges: int
    graph_sizes: Optional[Tensor] = None      # HOST int64 [2, B] nodes / in-edges per graph, when some graph lies beyond a graph tile
                                              # (ops.GraphPlan.build's hint: the list of such graphs is then made without a sync)

    def to(self, device, non_blocking: bool = True) -> "SceneGraphBatch":
        mv = lambda t: t.to(device, non_blocking=non_blocking)
        return SceneGraphBatch(mv(self.x), mv(self.edge_index), mv(self.edge_attr), mv(self.x_bbox),
                               mv(self.added_sym_edge), mv(self.batch), mv(self.ptr), self.num_graphs,
                               self.max_nodes, self.max_edges, self.graph_sizes)


class BatchBuffers:
    """Reusable (pinned) host buffers for one in-flight batch: grown to the high-water mark, so a steady-state collate
    neither allocates nor page-faults.  Keep one per batch in flight (the async H2D copy reads from it)."""

    def __init__(self, pin_memory: bool = False):
        self.pin = pin_memory
        self._buf = {}

    def _get(self, name: str, numel: int) -> Tensor:
        t = self._buf.get(name)
        if t is None or t.numel() < numel:
            t = torch.empty(max(numel, 1) * 5 // 4 + 16, dtype=torch.int64, pin_memory=self.pin)
            self._buf[name] = t
        return t[:numel]

    def views(self, N: int, E: int, S: int, B: int):
        return (self._get("x", N * 4).view(N, 4), self._get("edge_index", 2 * E).view(2, E), self._get("edge_attr", E),
                self._get("x_bbox", N * 4).view(N, 4), self._get("added_sym_edge", S), self._get("batch", N),
                self._get("ptr", B + 1))


class SceneGraphStore:
    """All scene graphs of one or more GQA JSON files, converted once (GQASceneGraphs, scene_graph.py:49-72)."""

    def __init__(self, vocab: SceneGraphVocab):
        lib = load()
        self.vocab = vocab
        self._h = c_void_p()
        _check(lib.isg_sg_store_create(vocab._h, ctypes.byref(self._h)), "isg_sg_store_create")

    def add_json_file(self, path: str) -> "SceneGraphStore":
        _check(load().isg_sg_store_add_json_file(self._h, os.fsencode(path)), f"isg_sg_store_add_json_file({path})")
        return self

    def add_json(self, text) -> "SceneGraphStore":
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        _check(load().isg_sg_store_add_json(self._h, data, len(data)), "isg_sg_store_add_json")
        return self

    def __len__(self) -> int:
        return int(load().isg_sg_store_num_graphs(self._h))

    def __contains__(self, image_id: str) -> bool:
        return load().isg_sg_store_find(self._h, image_id.encode("utf-8")) >= 0

    def slots(self, image_ids: Sequence[str]) -> Tensor:
        """Store slot of every id (-1 = unknown -> dummy graph); resolve once per dataset, collate by slot per batch."""
        out = torch.empty(len(image_ids), dtype=torch.int64)
        _check(load().isg_sg_store_find_many(self._h, _c_strings(image_ids), len(image_ids), out.data_ptr()),
               "isg_sg_store_find_many")
        return out

    def collate(self, image_ids, pin_memory: Optional[bool] = None, out: Optional["BatchBuffers"] = None,
                threads: int = 4) -> SceneGraphBatch:
        """query_and_translate per graph + Batch.from_data_list, written into (pinned) host tensors.
        ``image_ids``: id strings, or an int64 tensor of slots from ``slots()``.  ``out``: reusable buffers."""
        lib = load()
        slots = image_ids if torch.is_tensor(image_ids) else self.slots(image_ids)
        slots = slots.contiguous()
        B = slots.numel()
        tot = (c_int64 * 3)()
        _check(lib.isg_sg_collate_sizes(self._h, slots.data_ptr(), B, tot), "isg_sg_collate_sizes")
        N, E, S = int(tot[0]), int(tot[1]), int(tot[2])
        if out is None:
            pin = torch.cuda.is_available() if pin_memory is None else pin_memory
            out = BatchBuffers(pin)
        x, ei, ea, bb, sym, batch, ptr = out.views(N, E, S, B)
        bounds = (c_int64 * 2)()
        _check(lib.isg_sg_collate(self._h, slots.data_ptr(), B, x.data_ptr(), ei.data_ptr(), ea.data_ptr(),
                                  bb.data_ptr(), sym.data_ptr(), batch.data_ptr(), ptr.data_ptr(), bounds,
                                  int(threads)), "isg_sg_collate")
        sizes = None
        if int(bounds[0]) > 64 or int(bounds[1]) > 256:      # a graph beyond the 64-node / 256-in-edge tiles of the kernels
            sizes = torch.stack([ptr[1:] - ptr[:-1], torch.bincount(batch[ei[1]], minlength=B)]) if E > 0 else \
                torch.stack([ptr[1:] - ptr[:-1], torch.zeros(B, dtype=torch.int64)])
        return SceneGraphBatch(x, ei, ea, bb, sym, batch, ptr, B, int(bounds[0]), int(bounds[1]), sizes)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.isg_sg_store_free(self._h)
            self._h = None


def read_token_lists(meta_info_dir: str) -> List[List[str]]:
    """The six vocabulary sources in the reference's order (scene_graph.py:152-163)."""
    import json

    def lines(name):
        with open(os.path.join(meta_info_dir, name)) as f:
            return f.read().splitlines()

    def js(name):
        with open(os.path.join(meta_info_dir, name)) as f:
            return json.load(f)
    return [lines("name_gqa.txt"), lines("attr_gqa.txt"), lines("rel_gqa.txt"), js("objects.json"),
            js("predicates.json"), js("attributes.json")]


def declared_symbols(header_path: str) -> List[str]:
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(isg_[a-z0-9_]+)\s*\(", text)))
