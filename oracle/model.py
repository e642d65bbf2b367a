"""CPU restatement of the ISubGVQA inference hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional style: every function takes a ``sd`` mapping with the reference's
``state_dict`` key layout plus a key prefix, so the same
state_dict drives the product modules and this oracle.  All arithmetic is
fp32 torch-CPU, op-for-op in the reference's (unfused, PyG-style) order:
row gathers -> elementwise chain -> segment softmax -> scatter-add.

Every reference quirk listed in SURVEY Appendix B is reproduced on purpose
(Q1 pads compete in top-k, Q2 Gumbel noise in eval + relaxed top-k, Q3 double
[batch] indexing, Q4 .view scramble, Q5 float padding mask, Q6 un-offset
added_sym_edge, Q8 masked edges keep softmax mass, Q10 fp64 GraphNorm).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import primitives as P
from . import samplers as S


@dataclass
class PathConfig:
    """Flags that reach the hot path."""
    heads: int = 4
    negative_slope: float = 0.2
    masking_thresholds: List[float] = field(default_factory=lambda: [1.0, 1.0, 1.0, 0.15])
    use_topk: bool = True
    sampler_type: str = "gumbel"          # gumbel | imle | aimle
    sample_k: int = 5
    tau: float = 1.0                      # --tau (aimle theta temperature, masking.py:275)
    gumbel_tau: float = 0.1               # GumbelSampler default (gumbel_scheme.py:15)
    interpretable_mode: bool = False
    use_global_mask: bool = False
    # BASELINE configs[4] "fp16 features / fp32 accumulate": the projected rows x_l / x_r / e_proj and the aggregated
    # output (+bias) are rounded to IEEE half once each; all arithmetic stays fp32
    fp16_features: bool = False
    # training mode: train samplers + custom backward rules; dropout is NOT modelled (treated as p=0)
    training: bool = False
    imle_alpha: float = 1.0
    imle_beta: float = 10.0
    aimle_states: Optional[dict] = None   # layer prefix -> samplers.AimleTargetState (stateful, one per conv)
    graphnorm_eps: float = 1e-5
    nhead_text: int = 8


def linear(sd: Mapping[str, Tensor], p: str, x: Tensor) -> Tensor:
    w = sd[p + ".weight"]
    b = sd.get(p + ".bias") if hasattr(sd, "get") else None
    return F.linear(x, w, b)


# ---------------------------------------------------------------------------
# A8  NodeMaskToEdgeMask.forward (sampling/node_edge_masks.py:7-10)
# ---------------------------------------------------------------------------
class _NodeMaskToEdgeMaskFn(torch.autograd.Function):
    """The reference's custom rule (node_edge_masks.py:13-19): the edge-mask gradient goes to the destination only."""

    @staticmethod
    def forward(ctx, mask, edge_index):
        ctx.save_for_backward(edge_index)
        ctx.n = mask.shape[0]
        return (mask[edge_index[0]] * mask[edge_index[1]]).to(torch.float)

    @staticmethod
    def backward(ctx, grad_output):
        (edge_index,) = ctx.saved_tensors
        return P.scatter_sum(grad_output, edge_index[1], ctx.n), None


def node_mask_to_edge_mask(mask: Tensor, edge_index: Tensor) -> Tensor:
    if mask.requires_grad:
        return _NodeMaskToEdgeMaskFn.apply(mask, edge_index)
    return (mask[edge_index[0]] * mask[edge_index[1]]).to(torch.float)


# ---------------------------------------------------------------------------
# A7  MaskingModel.forward (models/masking.py:132-199), use_all_instrs=False
# ---------------------------------------------------------------------------
def node_gate_scores(sd, p: str, x: Tensor, u: Tensor, batch: Tensor) -> Tensor:
    """masking.py:137,151-155: gelu(<node_nn(x)_n, ques_nn(u)[batch]_n>/sqrt(C)) -> [N,1].

    ``u`` is whatever the caller passes; MaskingGATv2Conv passes imle_att[batch]
    (mgat_v2_conv.py:167) so ques_nn(u)[batch] reads row batch[batch[n]] (quirk Q3).
    """
    xn = P.gelu(linear(sd, p + ".node_nn.0", x))                       # :137
    q = P.gelu(linear(sd, p + ".ques_nn.0", u))[batch]                 # :152
    gate = torch.bmm(xn.unsqueeze(1), q.unsqueeze(2)).squeeze(-1) / torch.sqrt(
        torch.tensor(xn.size(1)))                                      # :151-153
    return P.gelu(gate)                                                # :155


def masking_model_forward(sd, p: str, x: Tensor, u: Tensor, batch: Tensor, cfg: PathConfig,
                          noise: Optional[Tensor] = None, return_aux: bool = False):
    gate = node_gate_scores(sd, p, x, u, batch)
    aux = {"gate": gate}
    if cfg.use_topk:                                                   # :158
        dense, m = P.to_dense_batch(gate, batch)                       # :162  (pad = 0.0, quirk Q1)
        aux["dense"] = dense
        if cfg.sampler_type == "imle" and cfg.training:                # :164-170 (sampler_train, masking.py:222-231)
            out = S.ImleTrain.apply(dense, noise, cfg.sample_k, cfg.imle_alpha, cfg.imle_beta, cfg.tau, cfg.tau)
            res = out.squeeze(0)[m]
        elif cfg.sampler_type == "aimle" and cfg.training:             # masking.py:257-268
            out = S.AimleTrain.apply(dense, noise, cfg.sample_k, cfg.aimle_states[p], cfg.tau, cfg.tau)
            res = out[m]
        elif cfg.sampler_type == "imle":                               # :163-173
            out = S.imle_eval(dense, cfg.sample_k, noise, 0.0)
            res = out.squeeze(0)[m]
        elif cfg.sampler_type == "aimle":
            out = S.aimle_eval(dense, cfg.sample_k, noise, cfg.tau)
            res = out[m]
        elif cfg.sampler_type == "simple":                             # :175-176; noise = the [1, B, n] torch.rand draw
            from . import simple as SS
            out, marg = SS.simple_forward(dense, cfg.sample_k, noise)
            aux["marginals"] = marg
            res = out.squeeze(0)[m]
        elif cfg.sampler_type == "gumbel":                             # :175-176
            out, khot, ind = S.gumbel_relaxed_topk(dense, cfg.sample_k, noise, cfg.gumbel_tau)
            aux["khot"], aux["ind"] = khot, ind
            res = out.squeeze(0)[m]
        else:
            raise NotImplementedError(cfg.sampler_type)
    else:                                                              # :195-198
        res = (torch.sigmoid(gate) > 0.5).to(gate.dtype)
    return (res, aux) if return_aux else res


# ---------------------------------------------------------------------------
# A6 + A6m  MaskingGATv2Conv.forward / message / aggregate
#           (models/mgat_v2_conv.py:138-279; PyG MessagePassing)
# ---------------------------------------------------------------------------
def gatv2_message_passing(x_l: Tensor, x_r: Tensor, e_proj: Tensor, att: Tensor,
                          edge_index: Tensor, edge_mask: Optional[Tensor],
                          negative_slope: float = 0.2):
    """The message-passing kernel boundary, unfused PyG order.

    x_l, x_r [N,H,C]; e_proj [E,H,C]; att [1,H,C]; edge_mask [E,1] or None.
    Returns (out [N,H,C], alpha [E,H]).
    """
    N = x_l.size(0)
    src, dst = edge_index[0], edge_index[1]
    x_j = x_l.index_select(0, src)
    x_i = x_r.index_select(0, dst)
    x = x_i + x_j                                                      # :253
    x = x + e_proj                                                     # :261
    if edge_mask is not None:
        x = x * edge_mask.unsqueeze(-1)                                # :264
    x = F.leaky_relu(x, negative_slope)                                # :266
    if edge_mask is not None:
        x = x * edge_mask.unsqueeze(-1)                                # :269
    alpha = (x * att).sum(dim=-1)                                      # :271
    alpha = P.pyg_softmax(alpha, dst, N)                               # :272
    if edge_mask is None:
        msg = x_j * alpha.unsqueeze(-1)                                # :278
    else:
        msg = x_j * (alpha * edge_mask).unsqueeze(-1)                  # :279
    out = P.scatter_sum(msg, dst, N)                                   # aggregate (aggr='add')
    return out, alpha


def gatv2_conv_forward(sd, p: str, x: Tensor, edge_index: Tensor, batch: Tensor,
                       edge_attr: Tensor, instruction: Tensor, imle_att: Tensor,
                       masking_threshold: float, cfg: PathConfig,
                       noise: Optional[Tensor] = None, aux: Optional[dict] = None):
    H = cfg.heads
    HC = sd[p + ".lin_l.weight"].shape[0]
    C = HC // H
    x = x * instruction[batch]                                         # :156
    x = P.gelu(x)                                                      # :157
    mask = None
    edge_mask = None
    thr = int(masking_threshold) if masking_threshold > 1 else masking_threshold   # masking.py:70-72
    if thr != 1.0:                                                     # :161
        r = masking_model_forward(sd, p + ".mask", x, imle_att[batch], batch, cfg, noise,
                                  return_aux=aux is not None)          # :166-168
        if aux is not None:
            mask, a = r
            aux.update(a)
        else:
            mask = r
        edge_mask = node_mask_to_edge_mask(mask, edge_index)           # :169-171
    x_l = linear(sd, p + ".lin_l", x).view(-1, H, C)                   # :177
    x_r = linear(sd, p + ".lin_r", x).view(-1, H, C)                   # :181
    e_proj = F.linear(edge_attr, sd[p + ".lin_edge.weight"]).view(-1, H, C)        # :259-260
    if cfg.fp16_features:
        x_l, x_r, e_proj = x_l.half().float(), x_r.half().float(), e_proj.half().float()
    out, alpha = gatv2_message_passing(x_l, x_r, e_proj, sd[p + ".att"], edge_index,
                                       edge_mask, cfg.negative_slope)
    out = out.view(-1, H * C)                                          # :227
    out = out + sd[p + ".bias"]                                        # :232
    if cfg.fp16_features:
        out = out.half().float()
    return out, mask, alpha


# ---------------------------------------------------------------------------
# A9  scatter_scaled_dot_product_attention (utils/scatter_scaled_dot_product.py:6-15)
# ---------------------------------------------------------------------------
def scatter_scaled_dot_product_attention(query: Tensor, key: Tensor, value: Tensor, batch: Tensor,
                                         num_graphs: Optional[int] = None) -> Tensor:
    B = query.size(0) if num_graphs is None else num_graphs
    logits = torch.bmm(query[batch].unsqueeze(1), key.unsqueeze(1).transpose(-2, -1)).squeeze() \
        / math.sqrt(query.size(-1))
    logits = logits.reshape(-1)
    att = P.scatter_softmax_1d(logits, batch, B)
    return att.unsqueeze(1) * value


# ---------------------------------------------------------------------------
# A5  MGAT.forward (models/mgat.py:110-184)
# ---------------------------------------------------------------------------
def mgat_forward(sd, p: str, x: Tensor, edge_index: Tensor, instr_vectors: Tensor,
                 global_language_feats: Tensor, edge_attr: Tensor, batch: Tensor,
                 cfg: PathConfig, noises: Optional[Dict[int, Tensor]] = None,
                 trace: Optional[list] = None):
    h = x
    mask = None
    L = len(cfg.masking_thresholds)
    global_mask = torch.ones((h.size(0), 1)) if cfg.use_global_mask else None
    for i in range(L):
        ins = instr_vectors[i]
        aux = {} if trace is not None else None
        conv_res, mask, alpha = gatv2_conv_forward(
            sd, f"{p}.convs.{i}", h, edge_index, batch, edge_attr, ins, global_language_feats,
            cfg.masking_thresholds[i], cfg, None if noises is None else noises.get(i), aux)
        conv_out = conv_res
        conv_res = P.gelu(linear(sd, f"{p}.x_proj.{i}.0", conv_res))   # :156
        conv_res = P.gelu(linear(sd, f"{p}.x_proj.{i}.2", conv_res))
        if cfg.use_global_mask:
            global_mask = mask * global_mask                           # :161-162
        conv_res = scatter_scaled_dot_product_attention(ins, conv_res, conv_res, batch)   # :168
        conv_res = P.graph_norm(conv_res, batch, sd[f"{p}.bns.{i}.weight"], sd[f"{p}.bns.{i}.bias"],
                                sd[f"{p}.bns.{i}.mean_scale"], cfg.graphnorm_eps,
                                num_graphs=instr_vectors.size(1))     # :171
        h = conv_res + h                                               # :172
        if cfg.use_global_mask:
            h = global_mask * h                                        # :174-175
        elif cfg.interpretable_mode and mask is not None:
            h = mask * h                                               # :176-177
        if trace is not None:
            trace.append({"conv_out": conv_out, "mask": mask, "alpha": alpha, "h": h, **aux})
    return h, mask


# ---------------------------------------------------------------------------
# A11  GlobalAttention.forward (models/att_pooling.py:57-77)
# ---------------------------------------------------------------------------
def global_attention_forward(sd, p: str, x: Tensor, u: Tensor, batch: Tensor,
                             node_mask: Optional[Tensor] = None, size: Optional[int] = None):
    B = int(batch[-1]) + 1 if size is None else size                   # :60
    x = linear(sd, p + ".node_nn.2", P.gelu(linear(sd, p + ".node_nn.0", x)))        # :62
    if node_mask is not None:
        x = x * node_mask                                              # :64
    q = linear(sd, p + ".ques_nn.2", P.gelu(linear(sd, p + ".ques_nn.0", u)))
    gate = torch.bmm(x.unsqueeze(1), q[batch].unsqueeze(2)).squeeze(-1) / torch.sqrt(
        torch.tensor(x.size(1)))                                       # :66-68
    gate = P.pyg_softmax(gate, batch, B)                               # :71
    out = P.scatter_sum(gate * x, batch, B)                            # :73
    return out, gate


# ---------------------------------------------------------------------------
# A2/A3  QuestionEncoder / QuestionDecoder (torch.nn.Transformer* post-norm, ReLU)
# ---------------------------------------------------------------------------
def _layer_norm(sd, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.size(-1),), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mha(sd, p: str, q_in: Tensor, kv_in: Tensor, nhead: int,
         key_bias: Optional[Tensor] = None) -> Tensor:
    """nn.MultiheadAttention forward, batch_first=False: inputs [T,B,D].

    key_bias [B,S] float is ADDED to the attention scores (quirk Q5: a float
    src_key_padding_mask is an additive bias).
    """
    T, B, D = q_in.shape
    Skv = kv_in.size(0)
    hd = D // nhead
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, w[:D], b[:D])
    k = F.linear(kv_in, w[D:2 * D], b[D:2 * D])
    v = F.linear(kv_in, w[2 * D:], b[2 * D:])
    q = q.reshape(T, B * nhead, hd).transpose(0, 1)
    k = k.reshape(Skv, B * nhead, hd).transpose(0, 1)
    v = v.reshape(Skv, B * nhead, hd).transpose(0, 1)
    scores = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(hd)
    if key_bias is not None:
        kb = key_bias.view(B, 1, 1, Skv).expand(-1, nhead, -1, -1).reshape(B * nhead, 1, Skv)
        scores = scores + kb
    attn = torch.softmax(scores, dim=-1)
    out = torch.bmm(attn, v).transpose(0, 1).reshape(T, B, D)
    return linear(sd, p + ".out_proj", out)


def clip_text_embeddings(sd, p: str, input_ids: Tensor) -> Tensor:
    """CLIPTextEmbeddings.forward: token_embedding(ids) + position_embedding(arange(T)) (isubgvqa.py:119-120)."""
    T = input_ids.size(1)
    tok = sd[p + ".token_embedding.weight"][input_ids]
    pos = sd[p + ".position_embedding.weight"][:T]
    return tok + pos.unsqueeze(0)


def question_encoder_forward(sd, p: str, questions: Tensor, mask: Tensor, nhead: int = 8,
                             nlayers: Optional[int] = None) -> Tensor:
    """QuestionEncoder.forward (models/question_encoder.py:28-38) -> [T,B,D]."""
    src = clip_text_embeddings(sd, p + ".text_vocab_embedding", questions)        # :32
    x = src.permute(1, 0, 2)                                                      # :36
    kb = mask.float()                                                             # :36 (additive!)
    i = 0
    while f"{p}.transformer_encoder.layers.{i}.linear1.weight" in sd and (nlayers is None or i < nlayers):
        lp = f"{p}.transformer_encoder.layers.{i}"
        x = _layer_norm(sd, lp + ".norm1", x + _mha(sd, lp + ".self_attn", x, x, nhead, kb))
        ff = linear(sd, lp + ".linear2", F.relu(linear(sd, lp + ".linear1", x)))
        x = _layer_norm(sd, lp + ".norm2", x + ff)
        i += 1
    return _layer_norm(sd, p + ".transformer_encoder.norm", x)


def question_decoder_forward(sd, p: str, memory: Tensor, nhead: int = 8) -> Tensor:
    """QuestionDecoder.forward (models/question_decoder.py:54-71) -> [n_ins,B,D]."""
    B = memory.size(1)
    x = sd[p + ".query_embed.weight"].unsqueeze(1).repeat(1, B, 1)                # :61-63
    i = 0
    while f"{p}.coarse_decoder.layers.{i}.linear1.weight" in sd:
        lp = f"{p}.coarse_decoder.layers.{i}"
        x = _layer_norm(sd, lp + ".norm1", x + _mha(sd, lp + ".self_attn", x, x, nhead))
        x = _layer_norm(sd, lp + ".norm2", x + _mha(sd, lp + ".multihead_attn", x, memory, nhead))
        ff = linear(sd, lp + ".linear2", F.relu(linear(sd, lp + ".linear1", x)))
        x = _layer_norm(sd, lp + ".norm3", x + ff)
        i += 1
    return _layer_norm(sd, p + ".coarse_decoder.norm", x)


# ---------------------------------------------------------------------------
# A4  SceneGraphEncoder.forward (models/scene_graph_encoder.py:53-143), eval mode
# ---------------------------------------------------------------------------
def _batchnorm_eval(sd, p: str, x: Tensor, eps: float = 1e-5, training: bool = False) -> Tensor:
    if training:      # train(): batch statistics (the running-stat update is a side effect, not modelled)
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, eps)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def scene_graph_encoder_forward(sd, p: str, x: Tensor, edge_index: Tensor, edge_attr: Tensor,
                                batch: Tensor, x_bbox: Tensor, added_sym_edge: Tensor,
                                cfg: PathConfig):
    emb = sd[p + ".sg_vocab_embedding.weight"]
    # nn.Embedding(..., padding_idx=<pad>) (scene_graph_encoder.py:19-21): same values, no gradient into the pad row
    x_embed_sum = F.embedding(x, emb, padding_idx=1).sum(dim=-2)                   # :63-70
    xb = x_bbox.to(x_embed_sum.dtype)                                              # :72
    tr = cfg.training
    xb = P.gelu(linear(sd, p + ".bbox_encoding.1", _batchnorm_eval(sd, p + ".bbox_encoding.0", xb, training=tr)))
    xb = P.gelu(linear(sd, p + ".bbox_encoding.4", _batchnorm_eval(sd, p + ".bbox_encoding.3", xb, training=tr)))
    xs = torch.cat((x_embed_sum, xb), dim=1)                                       # :73
    xs = P.gelu(linear(sd, p + ".feat_reduc.1", _batchnorm_eval(sd, p + ".feat_reduc.0", xs, training=tr)))   # :74
    e = F.embedding(edge_attr, emb, padding_idx=1).clone()                         # :76
    e[added_sym_edge, :] = e[added_sym_edge, :] * -1                               # :80 (quirk Q6)
    row, col = edge_index[0], edge_index[1]
    lp = p + ".scene_graph_encoding_layer"
    # MetaLayer: edge model first, then node model on the new edge features
    eo = torch.cat([xs[row], xs[col], e], 1)                                       # :119
    eo = linear(sd, lp + ".edge_model.edge_mlp.2", P.gelu(linear(sd, lp + ".edge_model.edge_mlp.0", eo)))
    no = torch.cat([xs[row], eo], dim=1)                                           # :139
    no = linear(sd, lp + ".node_model.node_mlp_1.2", P.gelu(linear(sd, lp + ".node_model.node_mlp_1.0", no)))
    no = P.scatter_mean(no, col, xs.size(0))                                       # :141
    no = torch.cat([xs, no], dim=1)                                                # :142
    xo = linear(sd, lp + ".node_model.node_mlp_2.2", P.gelu(linear(sd, lp + ".node_model.node_mlp_2.0", no)))
    # :99-102 GraphNorm in float64 (quirk Q10): fp32 params promote to fp64
    xo64 = xo.double()
    xn = P.graph_norm(xo64, batch, sd[p + ".graph_layer_norm.weight"], sd[p + ".graph_layer_norm.bias"],
                      sd[p + ".graph_layer_norm.mean_scale"], cfg.graphnorm_eps)
    return xn.float(), eo


# ---------------------------------------------------------------------------
# A1  ISubGVQA.forward (models/isubgvqa.py:213-297)
# ---------------------------------------------------------------------------
def language_features(sd, qst_feats: Tensor):
    """isubgvqa.py:244-247,265: the .view scramble (quirk Q4) + the two reductions."""
    flat = qst_feats.contiguous().view(qst_feats.size(1), int(qst_feats.size(0)),
                                       qst_feats.size(2)).flatten(1)               # :244-246
    glf = P.gelu(linear(sd, "qsts_reduction.0", flat))                             # :247
    instr = P.gelu(linear(sd, "instr_reduction.0", qst_feats))                     # :265
    return glf, instr


def classifier_head(sd, embed: Tensor, glf: Tensor) -> Tensor:
    feats = torch.cat((embed, glf, embed * glf), dim=1)                            # :288-290
    feats = P.gelu(linear(sd, "embedding.0", feats))                               # :291 (dropout off)
    return linear(sd, "logit_fc", feats)                                           # :292


def mgat_pool_classify(sd, x: Tensor, edge_index: Tensor, edge_attr: Tensor, batch: Tensor,
                       instr: Tensor, glf: Tensor, cfg: PathConfig,
                       noises: Optional[Dict[int, Tensor]] = None, trace: Optional[list] = None):
    """isubgvqa.py:267-292: MGAT -> GlobalAttention -> classifier (BASELINE config-2 workload)."""
    h, mask = mgat_forward(sd, "gat_seq", x, edge_index, instr[:4], glf, edge_attr, batch, cfg,
                           noises, trace)
    embed, gate = global_attention_forward(sd, "graph_global_attention_pooling", h, glf, batch,
                                           node_mask=mask, size=glf.size(0))
    logits = classifier_head(sd, embed, glf)
    return logits, mask, gate


def isubgvqa_forward(sd, node_embeddings: Tensor, edge_index: Tensor, edge_embeddings: Tensor,
                     batch: Tensor, questions: Tensor, qsts_att_mask: Tensor, x_bbox: Tensor,
                     added_sym_edge: Tensor, cfg: PathConfig,
                     noises: Optional[Dict[int, Tensor]] = None, text_uniform: Optional[Tensor] = None,
                     trace: Optional[list] = None):
    enc = question_encoder_forward(sd, "question_encoder", questions, qsts_att_mask, cfg.nhead_text)  # :228
    mask_text = None
    if text_uniform is not None:                                       # --text_sampling (:229-241), k = mgat_layers
        from . import simple as SS
        keys = P.gelu(linear(sd, "qsts_att_keys.0", enc))
        queries = P.gelu(linear(sd, "qsts_att_query.0", enc))
        logits = torch.bmm(keys.permute(1, 0, 2), queries.permute(1, 2, 0)).sum(-1) / math.sqrt(enc.size(-1))
        mask_text, _ = SS.simple_forward(logits.unsqueeze(-1), len(cfg.masking_thresholds), text_uniform)
        enc = (enc.permute(1, 0, 2) * mask_text.squeeze(0)).permute(1, 0, 2)
    qst_feats = question_decoder_forward(sd, "program_decoder", enc, cfg.nhead_text)                 # :243
    glf, instr = language_features(sd, qst_feats)
    x_enc, e_enc = scene_graph_encoder_forward(sd, "scene_graph_encoder", node_embeddings, edge_index,
                                               edge_embeddings, batch, x_bbox, added_sym_edge, cfg)  # :255
    logits, mask, gate = mgat_pool_classify(sd, x_enc, edge_index, e_enc, batch, instr, glf, cfg, noises, trace)
    return logits, mask, gate, [], mask_text                                                         # :297
