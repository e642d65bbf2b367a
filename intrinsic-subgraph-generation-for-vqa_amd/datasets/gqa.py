"""Batch assembly in front of ISubGVQA.forward.

Reference behaviour: gqa_collate, ISubGVQA/datasets/gqa.py:237-272: zip the dataset items, tokenise the question strings
with the CLIP tokenizer (padding=True), Batch.from_data_list the scene graphs, LongTensor the labels.  Here the dataset
item carries the IMAGE ID instead of a converted graph (the conversion happened once, inside the C++ store) and the
scene-graph part of the batch is one isg_sg_collate call into pinned memory.  The tokenizer is injected: the reference's
is a download (isubgvqa.py:116-119, CLIPTokenizer.from_pretrained) and stays outside this path.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch


def gqa_collate(data: Sequence[tuple], scene_graphs, tokenizer: Optional[Callable] = None, pin_memory: Optional[bool] = None):
    """data: (questionID, image_id, question_text, qst_types, short_answer_label, image_id) tuples (gqa.py:186-193 with the
    scene graph replaced by its image id).  Returns the reference's 7-tuple (gqa.py:262-270)."""
    question_id, graph_ids, question_text, qst_types, labels, img = zip(*data)
    qsts_padded = qsts_attention_mask = None
    if tokenizer is not None:
        q = tokenizer(list(question_text), return_tensors="pt", padding=True)            # :252-256
        qsts_padded, qsts_attention_mask = q["input_ids"], q["attention_mask"]
    scene_graph = scene_graphs.collate(list(graph_ids), pin_memory)                      # :258
    return (question_id, scene_graph, qsts_padded, qsts_attention_mask, torch.LongTensor(labels), img, qst_types)
