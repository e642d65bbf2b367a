"""Module path kept for drop-in imports (reference: ISubGVQA/sampling/methods/aimle.py)."""
from .perturb_and_map import aimle  # noqa: F401
