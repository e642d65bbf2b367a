"""Module path kept for drop-in imports (reference: ISubGVQA/models/positional_encoding.py)."""
from .text_encoder import PositionalEncoding  # noqa: F401
