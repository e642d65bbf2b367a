"""Model factory with the reference's entry point: ``build_model(args, cfg)`` (ISubGVQA/models/build.py:4-27).

The flags that select the hot path's variants are read off the argparse namespace by name and handed to the model as
keywords; I-MLE style sampling is always on, as in the reference."""
from .isubgvqa import ISubGVQA

# namespace attribute -> ISubGVQA keyword (identical names; listed so that a missing flag fails here, by name)
_FORWARDED = ("use_masking", "use_instruction", "use_mgat", "mgat_masks", "use_topk", "interpretable_mode",
              "concat_instr", "embed_cat")


def build_model(args, cfg=None):
    missing = [name for name in _FORWARDED if not hasattr(args, name)]
    if missing:
        raise AttributeError(f"build_model: args lacks {missing} (see ISubGVQA/utils/arg_parser.py)")
    kwargs = {name: getattr(args, name) for name in _FORWARDED}
    net = ISubGVQA(args, use_imle=True, **kwargs)
    return net.to(device=args.device)
