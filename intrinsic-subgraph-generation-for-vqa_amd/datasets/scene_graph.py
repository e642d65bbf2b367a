"""GQA scene graphs -> token tensors, on the C++ loader.

Reference behaviour: GQASceneGraphs, ISubGVQA/datasets/scene_graph.py:10-389.  The reference json.loads three files into
Python dicts (minutes and tens of GB for GQA) and converts an image's graph on every first access
(`query_and_translate` -> `convert_one_gqa_scene_graph`); here libisg_loader.so converts every image while it parses the
file, once, into flat int arrays, and `query_and_translate` / `collate` copy slices of them.

Differences, on purpose:
  * paths are constructor arguments (defaults = the reference's hard-coded ./ISubGVQA/... locations);
  * no GloVe download: `vectors` is only built when a `glove` mapping {token: Tensor[300]} is supplied;
  * attribute slots follow first-occurrence order (the reference iterates a Python set: hash-seed dependent).
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Optional, Sequence

import torch

from .. import loader

_SG_FILES = ("train_sceneGraphs.json", "val_sceneGraphs.json", "scene_graphs_test_dev.json")   # scene_graph.py:55-66


class GQASceneGraphs:
    def __init__(self, meta_info_dir: str = "./ISubGVQA/meta_info", scene_graph_dir: str = "./ISubGVQA/data/sceneGraphs",
                 scene_graph_files: Optional[Sequence[str]] = None, token_lists=None,
                 glove: Optional[Dict[str, torch.Tensor]] = None):
        if token_lists is None:
            token_lists = loader.read_token_lists(meta_info_dir)
        self.vocab_sg = loader.SceneGraphVocab(token_lists)
        print(f"Scene graph vocab size: {len(self.vocab_sg)}")                          # :52
        self.vectors = None
        if glove is not None:                                                            # :185-196
            stoi = self.vocab_sg.get_stoi()
            self.vectors = torch.randn(len(self.vocab_sg), 300)
            for tok, i in stoi.items():
                if tok in glove:
                    self.vectors[i] = glove[tok]
        self.store = loader.SceneGraphStore(self.vocab_sg)
        files = scene_graph_files if scene_graph_files is not None else [os.path.join(scene_graph_dir, f) for f in _SG_FILES]
        for path in files:                                                               # later files win, like dict `|` (:68-72)
            self.store.add_json_file(path)
        self.rel_mapping, self.obj_mapping, self.attr_mapping = {}, {}, {}              # :74-76 (always empty)

    def query_and_translate(self, queryID: str):
        """One image's graph with the reference's attribute names and shapes (x [n,4], edge_index [2,e], edge_attr [e,1],
        added_sym_edge [s], x_bbox [n,4]); unknown ids and single-edge graphs give the 6-node dummy (:71-143)."""
        b = self.store.collate([queryID], pin_memory=False)
        return SimpleNamespace(x=b.x, edge_index=b.edge_index, edge_attr=b.edge_attr.view(-1, 1),
                               added_sym_edge=b.added_sym_edge, x_bbox=b.x_bbox)

    def collate(self, image_ids: Sequence[str], pin_memory: Optional[bool] = None) -> loader.SceneGraphBatch:
        return self.store.collate(image_ids, pin_memory)
