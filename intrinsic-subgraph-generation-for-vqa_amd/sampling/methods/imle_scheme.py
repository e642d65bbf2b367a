"""Module path kept for drop-in imports (reference: ISubGVQA/sampling/methods/imle_scheme.py)."""
from .deterministic_scheme import IMLEScheme  # noqa: F401
