"""MI355X-native ISubGVQA inference hot path (import name: ``isubgvqa_amd``).

Layout: ``csrc/`` HIP kernels + C ABI (include/isg.h) -> ``_lib`` ctypes binding -> ``ops`` tensor
operators -> ``models`` / ``sampling`` / ``utils``: host-side mirror of the reference's
ISubGVQA/models, ISubGVQA/sampling and ISubGVQA/utils interfaces for this path.
"""
from . import ops  # noqa: F401
from .ops import GraphPlan  # noqa: F401

__all__ = ["ops", "GraphPlan"]
