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
Device work (inference): the torch modules only HOLD the parameters (so the state_dict keys are the reference's); the
forward pass below walks their layers itself: every projection and FFN layer is one launch of this library's bf16x6
matrix-core kernel with bias (+ReLU) fused (ops.linear: in_proj 512->1536 as ONE GEMM, out_proj, linear1+ReLU, linear2),
attention is isg_mha_small (csrc/isg_attn.hip: a question's K / V live in LDS, the float key-padding mask is the additive
bias it is in the reference), residual + LayerNorm are one launch of isg_add_layernorm.  Every producer leaves the row
maxima of what it wrote (ops.attach_row_maxima) so the fp16 three-product Linears downstream need no pass for their scales.  Training (autograd recording) runs the torch modules.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from .. import ops

FUSED_TEXT = True     # inference path on this library's kernels (A/B switch: False = torch's nn.Transformer* forward)


def _recording(module: torch.nn.Module, *inputs) -> bool:
    """True when the torch modules must run instead of the raw kernels (which have no autograd and no dropout): autograd is
    recording through a parameter OR an input (a frozen decoder over a trainable encoder's memory), or the module is in
    train() mode with a non-zero dropout somewhere in it."""
    if torch.is_grad_enabled() and (any(p.requires_grad for p in module.parameters()) or
                                    any(t is not None and t.requires_grad for t in inputs)):
        return True
    return module.training and any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in module.modules())


def _kernels_fit(T_kv: int, head_dim: int) -> bool:
    return ops.mha_small_supported(T_kv, head_dim)


def _attention(mha: torch.nn.MultiheadAttention, x_q: Tensor, x_kv: Tensor, B: int, key_bias=None, self_attn=True) -> Tensor:
    """nn.MultiheadAttention (batch_first=False) on flattened [T*B, D] rows: fused in_proj, isg_mha_small, out_proj."""
    D = x_q.size(1)
    w, b = mha.in_proj_weight, mha.in_proj_bias
    if self_attn:
        qkv = ops.linear(x_q, w, b)                                              # [T*B, 3D], one GEMM
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        wq = ops.derived_weight("mha_q", (w,), lambda: w[:D].contiguous())
        wkv = ops.derived_weight("mha_kv", (w,), lambda: w[D:].contiguous())
        bq = ops.derived_weight("mha_bq", (b,), lambda: b[:D].contiguous())
        bkv = ops.derived_weight("mha_bkv", (b,), lambda: b[D:].contiguous())
        q = ops.linear(x_q, wq, bq)
        kv = ops.linear(x_kv, wkv, bkv)                                          # [S*B, 2D]
        k, v = kv[:, :D], kv[:, D:]
    Tq, Tk, H = q.size(0) // B, k.size(0) // B, mha.num_heads
    if ops.mha_rows_supported(Tq, Tk, H, D // H) and ops.h3p_supported(q.size(0), mha.out_proj.weight.size(0), D):
        att = ops.mha_small(q, k, v, B, H, key_bias, planes_out=True)            # out_proj's operand leaves the kernel as planes32
    else:
        att = ops.mha_small(q, k, v, B, H, key_bias, want_rowmax=True)           # + max |att| per (row, head)
    return ops.linear(att, mha.out_proj.weight, mha.out_proj.bias)


def _ln(norm: torch.nn.LayerNorm, x: Tensor, residual: Tensor = None) -> Tensor:
    """LayerNorm(x + residual) as one launch; the result carries its row maxima, so none of the Linears that read it
    (in_proj, the cross-attention projections, linear1) makes a pass of its own for the fp16 planes' row scales."""
    return ops.add_layernorm(x, residual, norm)


def _ffn(layer, x: Tensor) -> Tensor:
    # ReLU is the layers' default activation; linear1's epilogue leaves one maximum per 32 columns for linear2's K-chunks
    w1, w2 = layer.linear1.weight, layer.linear2.weight
    if (ops.H3P_CHAIN and ops.h3p_supported(x.size(0), w1.size(0), w1.size(1)) and ops.h3p_supported(x.size(0), w2.size(0), w2.size(1))
            and w1.size(0) % 32 == 0):
        # the 2048-wide intermediate never exists as fp32 rows: linear1's epilogue writes it as the planes linear2 reads
        h = ops.linear_h3p(x, w1, layer.linear1.bias, relu=True, planes_out=True)
        return ops.linear_h3p(h, w2, layer.linear2.bias)
    h = ops.linear(x, w1, layer.linear1.bias, relu=True, want_rowmax=True)
    return ops.linear(h, w2, layer.linear2.bias)


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
        enc = self.transformer_encoder
        if (not FUSED_TEXT or not src.is_cuda or _recording(self, src)
                or not _kernels_fit(src.size(1), src.size(2) // enc.layers[0].self_attn.num_heads)):
            if src.is_cuda and not torch.is_grad_enabled():
                ops.COUNTERS["torch_attention"] += 1
            return enc(src.permute(1, 0, 2), src_key_padding_mask=mask.float())          # :35-37
        B, T, D = src.shape
        x = src.permute(1, 0, 2).reshape(T * B, D).contiguous()                          # torch's [T, B, D] row order
        key_bias = mask.float().contiguous()                                             # :36: ADDED to the scores (Q5)
        for layer in enc.layers:                                                         # post-norm encoder layers
            x = _ln(layer.norm1, x, _attention(layer.self_attn, x, x, B, key_bias))
            x = _ln(layer.norm2, x, _ffn(layer, x))
        out = _ln(enc.norm, x)
        return ops.carry_row_maxima(out.view(T, B, D), out)      # the decoder's cross-attention reads these rows


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
        dec = self.coarse_decoder
        if (not FUSED_TEXT or not memory.is_cuda or _recording(self, memory)
                or not _kernels_fit(max(memory.size(0), queries.size(0)), memory.size(2) // dec.layers[0].self_attn.num_heads)):
            if memory.is_cuda and not torch.is_grad_enabled():
                ops.COUNTERS["torch_attention"] += 1
            return dec(tgt=queries, memory=memory, tgt_mask=None)                        # :64-66
        S, _, D = memory.shape
        n = queries.size(0)
        x = queries.reshape(n * B, D).contiguous()
        mem = memory.reshape(S * B, D).contiguous()
        if mem.data_ptr() == memory.data_ptr():
            ops.carry_row_maxima(mem, memory)                                            # same rows, same order
        for layer in dec.layers:                                                         # post-norm decoder layers, no masks
            x = _ln(layer.norm1, x, _attention(layer.self_attn, x, x, B))
            x = _ln(layer.norm2, x, _attention(layer.multihead_attn, x, mem, B, self_attn=False))
            x = _ln(layer.norm3, x, _ffn(layer, x))
        return _ln(dec.norm, x).view(n, B, D)
