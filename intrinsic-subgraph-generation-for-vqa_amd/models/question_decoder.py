"""Module path kept for drop-in imports (reference: ISubGVQA/models/question_decoder.py)."""
from .text_encoder import QuestionDecoder  # noqa: F401
