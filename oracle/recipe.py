"""Seeded weight recipe for fixtures whose state_dict is too large to commit (TEST INFRASTRUCTURE ONLY).

The full ISubGVQA model at the reference's default architecture (C = 300, 4 MGAT layers, d = 512 text side) has ~36 M
parameters.  Golden G10 therefore stores inputs, outputs and a seed; the generating script (oracle/make_goldens.py,
running the real reference) and the tests both fill every state_dict entry with `fill_state_dict` below.  Every tensor
is drawn from its own generator seeded by (crc32(key) ^ seed), so the values do not depend on the iteration order or on
which extra keys a module carries; `checksums` lets a test detect a drifted RNG stream instead of failing mysteriously.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping

import torch
from torch import Tensor

_EMBEDDINGS = ("token_embedding.weight", "position_embedding.weight", "sg_vocab_embedding.weight", "query_embed.weight")


def canonical(key: str) -> str:
    """The CLIP embedding module is registered twice (isubgvqa.py:120,126-127: self.text_vocab_embedding and
    question_encoder.text_vocab_embedding share parameters): both names map to one recipe key."""
    key = key[7:] if key.startswith("module.") else key
    return key.replace("question_encoder.text_vocab_embedding.", "text_vocab_embedding.")


def recipe_tensor(key: str, like: Tensor, seed: int) -> Tensor:
    key = canonical(key)
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    shape = tuple(like.shape)
    if not like.is_floating_point():
        return torch.zeros(shape, dtype=like.dtype)                                   # num_batches_tracked
    if key.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if key.endswith("running_mean"):
        return 0.2 * torch.randn(shape, generator=g)
    if key.endswith("mean_scale") or (like.dim() == 1 and key.endswith("weight")):   # norm scales
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if like.dim() == 1:                                                               # biases
        return 0.1 * torch.randn(shape, generator=g)
    if key.endswith(_EMBEDDINGS):
        return 0.5 * torch.randn(shape, generator=g)
    return torch.randn(shape, generator=g) / math.sqrt(shape[-1])                     # Linear [out, in], att [1, H, C]


def fill_state_dict(module: torch.nn.Module, seed: int, skip=("pos_encoder.pe", "position_ids")) -> None:
    """Overwrite every parameter and buffer of `module` in place with the recipe."""
    with torch.no_grad():
        for key, t in module.state_dict().items():
            if any(s in key for s in skip):
                continue
            t.copy_(recipe_tensor(key, t, seed).to(t.dtype))


def checksums(sd: Mapping[str, Tensor], keys) -> Dict[str, float]:
    return {k: float(sd[k].double().sum()) for k in keys}
