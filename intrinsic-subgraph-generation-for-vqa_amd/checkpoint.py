"""Reader for the reference's training checkpoints.

The reference saves ``{"model": state_dict, "optimizer": ..., "lr_scheduler": ..., "epoch": int, "args": Namespace}``
with torch.save on rank 0 (ISubGVQA/training/train_loop.py:84-130, utils/misc.py:384-386); checkpoints written from a
DDP-wrapped model carry a ``module.`` prefix on every key (train_loop.py:89, main.py:87).  Its own loaders are
main.py:125-139 and run_token_coo.py:23-45 (``strict=True``).

``args`` is a pickled argparse.Namespace, so the file needs ``weights_only=False``: only load checkpoints you trust.
"""
from __future__ import annotations

import argparse
from typing import Any, Dict, Optional, Tuple

import torch

# flags the model constructor reads, with the reference's defaults (utils/arg_parser.py:13-116), used when an old
# checkpoint's Namespace lacks one of them
_ARG_DEFAULTS = dict(text_sampling=False, general_hidden_dim=300, distributed=False, mgat_layers=4, use_all_instrs=False,
                     use_global_mask=False, node_classification=False, sampler_type=None, sample_k=None, nb_samples=1,
                     alpha=1.0, beta=10.0, tau=1.0, use_masking=1, use_instruction=1, use_mgat=1,
                     mgat_masks=[1.0, 1.0, 1.0, 0.15], use_topk=True, interpretable_mode=False, concat_instr=0,
                     embed_cat=0, device="cuda")


def strip_ddp_prefix(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Drop the ``module.`` prefix DistributedDataParallel adds (only when every key has it)."""
    if state_dict and all(k.startswith("module.") for k in state_dict):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return dict(state_dict)


def read_checkpoint(path: str, map_location="cpu") -> Tuple[Dict[str, torch.Tensor], argparse.Namespace, Dict[str, Any]]:
    """-> (model state_dict without DDP prefix, args Namespace completed with defaults, remaining entries)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if "model" not in ckpt:
        raise KeyError(f"{path}: no 'model' entry (keys: {sorted(ckpt)})")
    sd = strip_ddp_prefix(ckpt["model"])
    args = ckpt.get("args")
    if args is None:
        args = argparse.Namespace()
    elif isinstance(args, dict):
        args = argparse.Namespace(**args)
    for k, v in _ARG_DEFAULTS.items():
        if not hasattr(args, k):
            setattr(args, k, v)
    rest = {k: v for k, v in ckpt.items() if k not in ("model", "args")}
    return sd, args, rest


def load_model(path: str, device: Optional[str] = None, strict: bool = True):
    """Build the drop-in ISubGVQA from a reference checkpoint and load its weights (strict by default, as
    run_token_coo.py:43 does).  Vocabulary sizes are taken from the checkpoint's embedding tables."""
    from .models.build import build_model
    sd, args, rest = read_checkpoint(path)
    args.sg_vocab_size = sd["scene_graph_encoder.sg_vocab_embedding.weight"].shape[0]
    args.text_vocab_size = sd["text_vocab_embedding.token_embedding.weight"].shape[0]
    if device is not None:
        args.device = device
    model = build_model(args, None)
    model.load_state_dict(sd, strict=strict)
    # the weight-plane caches are validated by (identity, tensor._version, data_ptr); under torch.inference_mode() tensors carry
    # no version counter, so weights written in place there would keep stale planes: a fresh load starts from empty caches
    from . import ops
    ops.invalidate_weight_cache()
    return model.eval(), args, rest
