"""Module path kept for drop-in imports (reference: ISubGVQA/models/question_encoder.py)."""
from .text_encoder import QuestionEncoder  # noqa: F401
