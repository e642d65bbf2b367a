"""CPU restatement of the host-side scene-graph conversion + collate.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, function by function:
  build_vocab            GQASceneGraphs.build_scene_graph_encoding_vocab, datasets/scene_graph.py:145-183, over
                         torchtext.vocab.vocab (torchtext 0.15/0.16 -- NOT vendored in /root/reference and not installed
                         here; its published algorithm is restated in `torchtext_vocab`)
  convert_scene_graph    GQASceneGraphs.convert_one_gqa_scene_graph, datasets/scene_graph.py:199-389
  query_and_translate    GQASceneGraphs.query_and_translate, datasets/scene_graph.py:71-143
  dataset_item           GQADataset.__getitem__ squeeze, datasets/gqa.py:170-175
  collate                torch_geometric.data.Batch.from_data_list (PyG 2.6, not installed) as used by gqa_collate,
                         datasets/gqa.py:258: `edge_index` is offset by the running node count and concatenated along
                         dim 1, every other attribute along dim 0 WITHOUT an offset -- including `added_sym_edge`
                         (quirk Q6) -- plus `batch` and `ptr`.

Pinned by tests/golden/g8_loader.pt: the per-graph tensors there come from the REAL convert_one_gqa_scene_graph /
query_and_translate (imported with stand-ins for torchtext / torch_geometric.data.Data, oracle/make_goldens.py).
One documented indeterminacy: the reference iterates `set(obj["attributes"])`, whose order depends on the interpreter's
string-hash seed; this restatement (and the C++ loader) use first-occurrence order, and the tests compare the attribute
slots of multi-attribute nodes as multisets.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

SPECIALS = ["<unk>", "<pad>", "<sos>", "<eos>", "<self>"]      # scene_graph.py:172-178
MAX_OBJ_TOKEN_LEN = 4                                          # :262 (1 name + 3 attributes)


def torchtext_vocab(ordered_dict: Dict[str, int], min_freq: int = 1, specials: Optional[Sequence[str]] = None,
                    special_first: bool = True) -> List[str]:
    """torchtext.vocab.vocab(): drop the specials from the dict, keep tokens whose VALUE is >= min_freq in dict order,
    put the specials in front.  Returns itos."""
    specials = list(specials or [])
    d = dict(ordered_dict)
    for tok in specials:
        d.pop(tok, None)
    tokens = [tok for tok, freq in d.items() if freq >= min_freq]
    return specials + tokens if special_first else tokens + specials


def build_vocab(token_lists: Sequence[Sequence[str]]) -> Dict[str, int]:
    """scene_graph.py:152-178.  The reference passes {token: position} where torchtext expects {token: frequency}: the
    token whose (last) position is 0 has 'frequency' 0 < min_freq and is dropped -- reproduced."""
    flat: List[str] = []
    for lst in token_lists:
        flat += list(lst)
    flat.append("<self>")
    flat.append("pokemon")
    stoi_pos = {tok: i for i, tok in enumerate(flat)}           # later duplicates overwrite the position, not the order
    itos = torchtext_vocab(stoi_pos, specials=SPECIALS)
    return {tok: i for i, tok in enumerate(itos)}


EMPTY_OBJECTS_SG = {"objects": {                               # scene_graph.py:205-229
    "0": {"name": "<unk>", "relations": [{"object": "1", "name": "<unk>"}], "attributes": ["<unk>"]},
    "1": {"name": "<unk>", "relations": [{"object": "0", "name": "<unk>"}], "attributes": ["<unk>"]},
}}
MISSING_SG = {"objects": {                                     # scene_graph.py:72-137
    str(i): {"name": "<unk>", "relations": [{"object": str(t), "name": "<unk>"}], "attributes": ["<unk>"]}
    for i, t in enumerate([1, 0, 3, 1, 5, 3])
}}


def convert_scene_graph(sg: dict, stoi: Dict[str, int]) -> Dict[str, torch.Tensor]:
    if len(sg["objects"]) == 0:
        sg = EMPTY_OBJECTS_SG
    obj_ids = sorted(sg["objects"].keys())                                              # :234
    node_of = {oid: i for i, oid in enumerate(obj_ids)}
    pairs = set()
    for i, oid in enumerate(obj_ids):                                                   # :255-263
        for rel in sg["objects"][oid]["relations"]:
            pairs.add((i, node_of[rel["object"]]))
    x, bbox, ei, ea, sym = [], [], [], [], []
    for i, oid in enumerate(obj_ids):
        obj = sg["objects"][oid]
        tok = [stoi["<pad>"]] * MAX_OBJ_TOKEN_LEN                                       # :281-283
        tok[0] = stoi.get(obj["name"], 1)                                               # :287 (OOV -> 1 = <pad>)
        seen = []
        for a in obj["attributes"]:                                                     # :294 set(): first occurrence here
            if a not in seen:
                seen.append(a)
        for j, a in enumerate(seen[:3]):                                                # :295-299
            tok[j + 1] = stoi.get(a, 1)
        x.append(tok)
        bbox.append([obj.get("x1", -1), obj.get("y1", -1), obj.get("x2", -1), obj.get("y2", -1)])   # :301-306
        ei.append([i, i])                                                               # :311 self loop first
        ea.append(stoi["<self>"])
        for rel in obj["relations"]:                                                    # :317-345
            j = node_of[rel["object"]]
            ei.append([i, j])
            t = stoi.get(rel["name"], 1)
            ea.append(t)
            if (j, i) not in pairs:                                                     # symmetric completion
                ei.append([j, i])
                ea.append(t)
                sym.append(len(ea) - 1)
    return dict(x=torch.tensor(x, dtype=torch.long), edge_index=torch.tensor(ei, dtype=torch.long).t().contiguous(),
                edge_attr=torch.tensor(ea, dtype=torch.long).view(-1, 1), x_bbox=torch.tensor(bbox),
                added_sym_edge=torch.tensor(sym, dtype=torch.long))


def query_and_translate(scene_graphs: Dict[str, dict], image_id: str, stoi: Dict[str, int]):
    d = convert_scene_graph(scene_graphs.get(image_id, MISSING_SG), stoi)               # :138-139
    if d["edge_index"].size(1) == 1:                                                    # :140-141
        d = convert_scene_graph(MISSING_SG, stoi)
    return d


def dataset_item(d: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """gqa.py:172-173: x.squeeze() (a no-op for n >= 2 nodes), edge_attr [E,1] -> [E]."""
    out = dict(d)
    out["x"] = d["x"].squeeze()
    out["edge_attr"] = d["edge_attr"].squeeze()
    return out


def collate(items: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    n = [int(it["x"].size(0)) for it in items]
    ptr = torch.zeros(len(items) + 1, dtype=torch.long)
    ptr[1:] = torch.tensor(n).cumsum(0)
    return dict(
        x=torch.cat([it["x"] for it in items], 0),
        edge_index=torch.cat([it["edge_index"] + int(ptr[g]) for g, it in enumerate(items)], 1),
        edge_attr=torch.cat([it["edge_attr"] for it in items], 0),
        x_bbox=torch.cat([it["x_bbox"] for it in items], 0),
        added_sym_edge=torch.cat([it["added_sym_edge"] for it in items], 0),            # NOT offset (quirk Q6)
        batch=torch.repeat_interleave(torch.arange(len(items)), torch.tensor(n)),
        ptr=ptr,
    )
