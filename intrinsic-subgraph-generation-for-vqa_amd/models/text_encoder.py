"""Dense language side of ISubGVQA: CLIP token embeddings, question encoder, program decoder.

Reference behaviour:
  * CLIPTextEmbeddings copied out of a CLIP model at ISubGVQA/models/isubgvqa.py:119-120
    (token_embedding [49408,512] + position_embedding [77,512]); here a same-key module with random
    init (no network: weights come from a checkpoint's state_dict).
  * QuestionEncoder, ISubGVQA/models/question_encoder.py:6-38: 4 post-norm TransformerEncoder layers
    (d=512, 8 heads, ReLU FFN 2048) + final LayerNorm; the HF attention_mask is cast to float and
    handed over as src_key_padding_mask, i.e. it is ADDED to the attention scores (+1 on real tokens,
    pads are attended: SURVEY App. B Q5).  emb_proj and the sinusoidal pos_encoder are constructed but
    not applied (:33-34).
  * QuestionDecoder, ISubGVQA/models/question_decoder.py:4-71: n_instructions learned queries through a
    3-layer post-norm TransformerDecoder over the encoder memory, no masks.
These are plain dense contractions: they run on torch's rocBLAS/hipBLASLt MFMA GEMMs.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor


class CLIPTextEmbeddings(torch.nn.Module):
    def __init__(self, vocab_size: int = 49408, hidden_size: int = 512, max_position_embeddings: int = 77):
        super().__init__()
        self.token_embedding = torch.nn.Embedding(vocab_size, hidden_size)
        self.position_embedding = torch.nn.Embedding(max_position_embeddings, hidden_size)
        self.register_buffer("position_ids", torch.arange(max_position_embeddings).unsqueeze(0), persistent=False)

    def forward(self, input_ids: Tensor) -> Tensor:
        T = input_ids.shape[-1]
        return self.token_embedding(input_ids) + self.position_embedding(self.position_ids[:, :T])


class PositionalEncoding(torch.nn.Module):
    """Sinusoidal table (ISubGVQA/models/positional_encoding.py); only its buffer key matters here."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = torch.nn.Dropout(p=dropout)
        pos = torch.arange(max_len, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * freq)
        pe[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", pe.unsqueeze(1))

    def forward(self, x):
        return self.dropout(x + self.pe[: x.size(0), :])


class QuestionEncoder(torch.nn.Module):
    def __init__(self, text_vocab_embedding, text_emb_dim, ninp, nhead, nhid, nlayers, dropout=0.5):
        super().__init__()
        self.text_vocab_embedding = text_vocab_embedding
        self.model_type = "Transformer"
        self.emb_proj = torch.nn.Linear(text_emb_dim, ninp)
        self.pos_encoder = PositionalEncoding(ninp, dropout)
        layer = torch.nn.TransformerEncoderLayer(ninp, nhead, nhid, dropout)
        self.transformer_encoder = torch.nn.TransformerEncoder(layer, nlayers, norm=torch.nn.LayerNorm(ninp),
                                                               enable_nested_tensor=False)
        self.ninp = ninp

    def forward(self, src: Tensor, mask: Tensor) -> Tensor:
        src = self.text_vocab_embedding(src)                                             # :32
        return self.transformer_encoder(src.permute(1, 0, 2), src_key_padding_mask=mask.float())   # :35-37


class QuestionDecoder(torch.nn.Module):
    def __init__(self, n_instructions, ninp, nhead, nhid, nlayers, dropout=0.1):
        super().__init__()
        self.model_type = "Transformer"
        self.num_queries = n_instructions
        self.query_embed = torch.nn.Embedding(self.num_queries, ninp)
        layer = torch.nn.TransformerDecoderLayer(ninp, nhead, nhid, dropout)
        self.coarse_decoder = torch.nn.TransformerDecoder(layer, nlayers, norm=torch.nn.LayerNorm(ninp))

    def forward(self, memory: Tensor) -> Tensor:
        B = memory.size(1)
        queries = self.query_embed.weight.unsqueeze(1).repeat(1, B, 1)                   # :61-63
        return self.coarse_decoder(tgt=queries, memory=memory, tgt_mask=None)            # :64-66
