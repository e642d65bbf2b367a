"""Module path kept for drop-in imports (reference: ISubGVQA/sampling/methods/wrapper.py)."""
from .perturb_and_map import imle  # noqa: F401
