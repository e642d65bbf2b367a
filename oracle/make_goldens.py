"""Generate golden vectors by running the REAL reference code from /root/reference.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (the reference never
travels to the GPU box); writes small data-only fixtures to tests/golden/*.pt:

  g1_gumbel.pt     GumbelSampler.forward           (reference module, direct import)
  g2_imle.pt       imle eval wrapper + IMLEScheme   (reference module, direct import)
  g3_aimle.pt      aimle eval wrapper               (reference module, direct import)
  g4_question.pt   QuestionEncoder + QuestionDecoder at reduced dims (direct import;
                   the CLIP embedding module is a same-key token+position stand-in because
                   the constructor would download weights, isubgvqa.py:119)
  g5_mgat_*.pt     MGAT / MaskingGATv2Conv / MaskingModel / GlobalAttention /
                   NodeMaskToEdgeMask / scatter attention: the reference's own glue code
                   executed over oracle/pyg_standin.py (restated third-party primitives)

  g6_sampler_grads.pt  training-mode samplers (GumbelSampler train=True straight-through; imle / aimle train wrappers
                   with their second MAP solve and the adaptive-beta state) -- forward outputs and input gradients
  g7_train_*.pt    MGAT + GlobalAttention in train() mode (dropout patched to identity), loss gradients for every
                   parameter and input, through the reference's custom backward rules

  g8_loader.pt     GQASceneGraphs.convert_one_gqa_scene_graph / query_and_translate (the real functions, imported over
                   stand-ins for torchtext and torch_geometric.data.Data) on synthetic scene-graph JSON with a synthetic
                   vocabulary: the JSON text, the token lists and the per-image tensors

  g9_simple.pt     EdgeSIMPLEBatched.forward (policy 'edge_candid'): masks, marginals and input gradients on ragged
                   padded batches, incl. rows with more zero pads than k (the circuit's padding accidents) and a NaN row

Usage:  TORCHDYNAMO_DISABLE=1 (the reference decorates its circuit passes with torch.compile) PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_goldens
No reference source text is written anywhere; fixtures hold tensors only.
"""
from __future__ import annotations

import os
import sys

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

from . import pyg_standin  # noqa: E402
from . import samplers as S  # noqa: E402


def _noise_from_seed(seed: int, shape, scale: float = 1.0) -> torch.Tensor:
    """The Gumbel draw the reference makes right after torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    return S.uniform_to_gumbel(torch.rand(shape), 0.0, scale)


def gen_samplers():
    from ISubGVQA.sampling.methods.gumbel_scheme import GumbelSampler
    from ISubGVQA.models.masking import get_imle_samplers, get_aimle_samplers

    g1 = []
    case = 0
    for (B, Nmax, k) in [(1, 8, 2), (4, 16, 5), (32, 20, 5), (4, 64, 5), (2, 200, 5), (3, 4, 5), (5, 37, 2)]:
        gen = torch.Generator().manual_seed(100 + case)
        scores = torch.randn(B, Nmax, 1, generator=gen)
        # zero-padded tails as to_dense_batch produces them (quirk Q1), gelu-like negatives included
        for b in range(B):
            n = int(torch.randint(1, Nmax + 1, (1,), generator=gen))
            if b == 0:
                n = Nmax
            scores[b, n:] = 0.0
        seed = 7000 + case
        sampler = GumbelSampler(k=k, policy="edge_candid", train_ensemble=1, val_ensemble=1)
        torch.manual_seed(seed)
        out, _ = sampler(scores.clone(), train=False)
        noise = _noise_from_seed(seed, (B, Nmax))
        g1.append(dict(scores=scores, k=k, seed=seed, noise=noise, out=out.clone()))
        case += 1
    torch.save(g1, os.path.join(OUT, "g1_gumbel.pt"))

    g2 = []
    case = 0
    for (B, Nmax, k) in [(1, 8, 2), (4, 16, 5), (8, 20, 5), (3, 4, 5), (3, 5, 5), (4, 12, 3)]:
        gen = torch.Generator().manual_seed(200 + case)
        scores = torch.randn(B, Nmax, 1, generator=gen)
        for b in range(B):
            n = int(torch.randint(1, Nmax + 1, (1,), generator=gen))
            scores[b, n:] = 0.0
        if B >= 3 and Nmax >= 8:
            scores[1, 2] = scores[1, 5] = scores[1].max() - 0.0   # exact ties at the top
            scores[2, :6] = 0.25                                   # ties that straddle the k-th value
        _, val = get_imle_samplers(sample_k=k, device="cpu", nb_samples=1, alpha=1.0, beta=10.0, tau=1.0)
        seed = 7100 + case
        torch.manual_seed(seed)
        out = val(scores.clone())
        g2.append(dict(scores=scores, k=k, seed=seed, out=out[0].clone()))
        case += 1
    torch.save(g2, os.path.join(OUT, "g2_imle.pt"))

    g3 = []
    case = 0
    for (B, Nmax, k, tau) in [(1, 8, 2, 1.0), (4, 16, 5, 1.0), (8, 20, 5, 1.0), (3, 4, 5, 1.0), (4, 33, 3, 0.5)]:
        gen = torch.Generator().manual_seed(300 + case)
        scores = torch.randn(B, Nmax, 1, generator=gen)
        for b in range(B):
            n = int(torch.randint(1, Nmax + 1, (1,), generator=gen))
            scores[b, n:] = 0.0
        _, val = get_aimle_samplers(sample_k=k, device="cpu", nb_samples=1, alpha=1.0, tau=tau)
        seed = 7200 + case
        torch.manual_seed(seed)
        out = val(scores.clone())
        noise = _noise_from_seed(seed, (B, 1, Nmax, 1), scale=0.3)
        g3.append(dict(scores=scores, k=k, tau=tau, seed=seed, noise=noise, out=out.clone()))
        case += 1
    torch.save(g3, os.path.join(OUT, "g3_aimle.pt"))


class _TextEmb(torch.nn.Module):
    """Same parameter keys and arithmetic as CLIPTextEmbeddings (token + position lookup)."""

    def __init__(self, vocab, dim, max_pos=77):
        super().__init__()
        self.token_embedding = torch.nn.Embedding(vocab, dim)
        self.position_embedding = torch.nn.Embedding(max_pos, dim)

    def forward(self, input_ids):
        T = input_ids.size(1)
        return self.token_embedding(input_ids) + self.position_embedding(torch.arange(T)).unsqueeze(0)


def gen_question():
    from ISubGVQA.models.question_encoder import QuestionEncoder
    from ISubGVQA.models.question_decoder import QuestionDecoder

    torch.manual_seed(11)
    D, nhead, nhid, vocab = 32, 4, 64, 50
    emb = _TextEmb(vocab, D)
    enc = QuestionEncoder(text_vocab_embedding=emb, text_emb_dim=D, ninp=D, nhead=nhead, nhid=nhid,
                          nlayers=2, dropout=0.1).eval()
    dec = QuestionDecoder(n_instructions=4, ninp=D, nhead=nhead, nhid=nhid, nlayers=2, dropout=0.1).eval()
    B, T = 5, 9
    q = torch.randint(0, vocab, (B, T))
    lens = torch.tensor([9, 4, 7, 1, 9])
    mask = (torch.arange(T)[None, :] < lens[:, None]).long()      # HF attention_mask: 1 = real token
    with torch.no_grad():
        e = enc(q, mask)
        d = dec(e)
    # pos_encoder.pe is a constructed-but-unused 5000-row buffer (question_encoder.py:33-34): not stored
    sd = {"question_encoder." + k: v.clone() for k, v in enc.state_dict().items() if "pos_encoder" not in k}
    sd.update({"program_decoder." + k: v.clone() for k, v in dec.state_dict().items()})
    torch.save(dict(sd=sd, questions=q, mask=mask, nhead=nhead, enc_out=e, dec_out=d),
               os.path.join(OUT, "g4_question.pt"))


def _graphs(kind: str, gen):
    """Tiny PyG-style batches.  Returns batch[N], edge_index[2,E] (unsorted, self-loops, duplicates)."""
    if kind == "edgecases":
        sizes = [3, 1, 5, 2]          # incl. a 1-node graph
    elif kind == "b1":
        sizes = [6]
    else:
        sizes = [int(torch.randint(2, 9, (1,), generator=gen)) for _ in range(6)]
    batch, src, dst = [], [], []
    off = 0
    for g, n in enumerate(sizes):
        batch += [g] * n
        for v in range(n):            # self loop per node first (scene_graph.py:309-343)
            if not (kind == "edgecases" and g == 2 and v == 4):   # node 4 of graph 2: isolated target
                src.append(off + v); dst.append(off + v)
        m = int(torch.randint(0, 2 * n + 1, (1,), generator=gen)) if n > 1 else 0
        for _ in range(m):
            a = int(torch.randint(0, n, (1,), generator=gen)); b = int(torch.randint(0, n, (1,), generator=gen))
            if kind == "edgecases" and g == 2 and b == 4:
                b = 0
            src.append(off + a); dst.append(off + b)
            if torch.rand(1, generator=gen) < 0.3:                # duplicate edge
                src.append(off + a); dst.append(off + b)
        off += n
    ei = torch.tensor([src, dst], dtype=torch.long)
    perm = torch.randperm(ei.size(1), generator=gen)              # edge order is arbitrary
    return torch.tensor(batch, dtype=torch.long), ei[:, perm]


def gen_mgat():
    from ISubGVQA.models.mgat import MGAT
    from ISubGVQA.models.att_pooling import GlobalAttention
    from ISubGVQA.models.masking import get_imle_samplers, get_aimle_samplers
    from ISubGVQA.sampling.node_edge_masks import NodeMaskToEdgeMask
    from ISubGVQA.utils.scatter_scaled_dot_product import scatter_scaled_dot_product_attention

    cases = [
        dict(name="gumbel_c8", C=8, L=3, masks=[1.0, 1.0, 0.15], sampler="gumbel", k=2, kind="rand", interp=False),
        dict(name="gumbel_edge", C=8, L=4, masks=[1.0, 0.15, 1.0, 0.15], sampler="gumbel", k=2, kind="edgecases", interp=False),
        dict(name="imle_c16", C=16, L=4, masks=[1.0, 1.0, 1.0, 0.15], sampler="imle", k=3, kind="rand", interp=False),
        dict(name="aimle_c12", C=12, L=2, masks=[0.15, 0.15], sampler="aimle", k=2, kind="rand", interp=True),
        dict(name="gumbel_b1", C=8, L=2, masks=[1.0, 0.15], sampler="gumbel", k=3, kind="b1", interp=True),
        dict(name="nomask_c20", C=20, L=2, masks=[1.0, 1.0], sampler="gumbel", k=2, kind="rand", interp=False),
    ]
    for ci, c in enumerate(cases):
        gen = torch.Generator().manual_seed(500 + ci)
        batch, ei = _graphs(c["kind"], gen)
        N, E, B, C, L = batch.numel(), ei.size(1), int(batch.max()) + 1, c["C"], c["L"]
        torch.manual_seed(40 + ci)
        model = MGAT(channels=C, num_ins=L, heads=4, use_instr=True, masking_thresholds=c["masks"],
                     use_topk=True, interpretable_mode=c["interp"], sampler_type=c["sampler"],
                     sample_k=c["k"], nb_samples=1, alpha=1.0, beta=10.0, tau=1.0).eval()
        pool = GlobalAttention(num_node_features=C, num_out_features=C).eval()
        with torch.no_grad():        # make biases / norm params non-trivial
            for n_, p_ in list(model.named_parameters()) + list(pool.named_parameters()):
                if n_.endswith("bias") or "bns" in n_:
                    p_.add_(0.1 * torch.randn(p_.shape, generator=gen))
        for conv in model.convs:     # the factories hard-code device="cuda" (masking.py:97,106; quirk Q7)
            if c["sampler"] == "imle":
                conv.mask.sampler_train, conv.mask.sampler_val = get_imle_samplers(
                    sample_k=c["k"], device="cpu", nb_samples=1, alpha=1.0, beta=10.0, tau=1.0)
            elif c["sampler"] == "aimle":
                conv.mask.sampler_train, conv.mask.sampler_val = get_aimle_samplers(
                    sample_k=c["k"], device="cpu", nb_samples=1, alpha=1.0, tau=1.0)
        x = torch.randn(N, C, generator=gen)
        edge_attr = torch.randn(E, C, generator=gen)
        instr = torch.randn(L, B, C, generator=gen)
        glf = torch.randn(B, C, generator=gen)
        seed = 9000 + ci
        torch.manual_seed(seed)
        with torch.no_grad():
            h, mask, _, _ = model(x=x, edge_index=ei, instr_vectors=instr, global_language_feats=glf,
                                  edge_attr=edge_attr, batch=batch, return_masks=True)
            emb, gate = pool(x=h, u=glf, batch=batch, size=None, return_mask=True, node_mask=mask)
        # replay the RNG stream the masked layers consumed, in order
        counts = torch.bincount(batch, minlength=B)
        nmax = int(counts.max())
        torch.manual_seed(seed)
        noises = {}
        for i, thr in enumerate(c["masks"]):
            if thr != 1.0:
                if c["sampler"] == "gumbel":
                    noises[i] = S.uniform_to_gumbel(torch.rand(B, nmax))
                else:
                    noises[i] = S.uniform_to_gumbel(torch.rand(B, 1, nmax, 1), 0.0, 0.3)
        # single conv layer (the last one) with its attention weights, for the kernel-level check
        li = L - 1
        torch.manual_seed(seed + 1)
        with torch.no_grad():
            conv_out, conv_mask, (_, alpha) = model.convs[li](
                x=x, edge_index=ei, edge_attr=edge_attr, instruction=instr[li], batch=batch,
                return_masks=True, return_attention_weights=True, imle_att=glf, all_instrs=instr)
            att9 = scatter_scaled_dot_product_attention(instr[li], x, x, batch)
            if conv_mask is not None:
                em = NodeMaskToEdgeMask.apply(conv_mask, ei, torch.tensor(N))
            else:
                em = None
        torch.manual_seed(seed + 1)
        conv_noise = None
        if c["masks"][li] != 1.0:
            conv_noise = (S.uniform_to_gumbel(torch.rand(B, nmax)) if c["sampler"] == "gumbel"
                          else S.uniform_to_gumbel(torch.rand(B, 1, nmax, 1), 0.0, 0.3))
        # node_logits.* (512x2577) is never used in forward (mgat.py:98-102): not stored
        sd = {"gat_seq." + k: v.clone() for k, v in model.state_dict().items() if "node_logits" not in k}
        sd.update({"graph_global_attention_pooling." + k: v.clone() for k, v in pool.state_dict().items()})
        torch.save(dict(cfg={k: c[k] for k in ("C", "L", "masks", "sampler", "k", "interp")}, sd=sd,
                        x=x, edge_index=ei, edge_attr=edge_attr, batch=batch, instr=instr, glf=glf,
                        noises=noises, h=h, mask=mask, pool_out=emb, pool_gate=gate,
                        conv_layer=li, conv_noise=conv_noise, conv_out=conv_out, conv_mask=conv_mask,
                        conv_alpha=alpha, conv_edge_mask=em, scatter_att=att9),
                   os.path.join(OUT, f"g5_mgat_{c['name']}.pt"))


def _padded_scores(B, Nmax, gen):
    scores = torch.randn(B, Nmax, 1, generator=gen)
    lens = []
    for b in range(B):
        n = int(torch.randint(1, Nmax + 1, (1,), generator=gen))
        if b == 0:
            n = Nmax
        scores[b, n:] = 0.0
        lens.append(n)
    return scores, torch.tensor(lens)


def gen_sampler_grads():
    """Training-mode samplers: outputs and gradients w.r.t. the dense scores for a random upstream gradient that is
    zero on the padded slots (what the `[mask]` gather of masking.py:170-176 sends back)."""
    from ISubGVQA.sampling.methods.gumbel_scheme import GumbelSampler
    from ISubGVQA.sampling.methods.aimle import aimle
    from ISubGVQA.sampling.methods.imle_scheme import IMLEScheme
    from ISubGVQA.sampling.methods.noise import GumbelDistribution
    from ISubGVQA.sampling.methods.target_aimle import AdaptiveTargetDistribution
    from ISubGVQA.models.masking import get_imle_samplers

    out = dict(gumbel=[], imle=[], aimle=[])
    for ci, (B, Nmax, k) in enumerate([(1, 8, 2), (4, 16, 5), (16, 20, 5), (3, 4, 5), (2, 130, 3)]):
        gen = torch.Generator().manual_seed(600 + ci)
        scores, lens = _padded_scores(B, Nmax, gen)
        w = torch.randn(B, Nmax, 1, generator=gen) * (torch.arange(Nmax)[None, :, None] < lens[:, None, None])
        seed = 7600 + ci
        sampler = GumbelSampler(k=k, policy="edge_candid", train_ensemble=1, val_ensemble=1)
        th = scores.clone().requires_grad_(True)
        torch.manual_seed(seed)
        res, _ = sampler(th, train=True)
        (res.squeeze(0) * w).sum().backward()
        out["gumbel"].append(dict(scores=scores, lens=lens, k=k, w=w, noise=_noise_from_seed(seed, (B, Nmax)),
                                  out=res.detach().clone(), grad=th.grad.clone()))
    for ci, (B, Nmax, k, beta) in enumerate([(1, 8, 2, 10.0), (4, 16, 5, 10.0), (16, 20, 5, 10.0), (3, 4, 5, 10.0),
                                             (6, 40, 3, 2.5)]):
        gen = torch.Generator().manual_seed(650 + ci)
        scores, lens = _padded_scores(B, Nmax, gen)
        w = 0.2 * torch.randn(B, Nmax, 1, generator=gen) * (torch.arange(Nmax)[None, :, None] < lens[:, None, None])
        seed = 7650 + ci
        train, _ = get_imle_samplers(sample_k=k, device="cpu", nb_samples=1, alpha=1.0, beta=beta, tau=1.0)
        th = scores.clone().requires_grad_(True)
        torch.manual_seed(seed)
        res = train(th)[0]
        (res.squeeze(0) * w).sum().backward()
        out["imle"].append(dict(scores=scores, lens=lens, k=k, beta=beta, w=w,
                                noise=_noise_from_seed(seed, (B, 1, Nmax, 1), 0.3),
                                out=res.detach().clone(), grad=th.grad.clone()))
    for ci, (B, Nmax, k, beta0, tau) in enumerate([(4, 16, 5, 0.0, 1.0), (8, 20, 5, 3.0, 1.0), (3, 12, 2, 0.7, 0.5)]):
        gen = torch.Generator().manual_seed(680 + ci)
        scheduler = IMLEScheme("edge_candid", k, 1, 1)
        target = AdaptiveTargetDistribution(initial_alpha=1.0, initial_beta=beta0)

        @aimle(target_distribution=target, noise_distribution=GumbelDistribution(0.0, 0.3, "cpu"), nb_samples=1,
               theta_noise_temperature=tau, target_noise_temperature=tau, symmetric_perturbation=True)
        def train(logits):
            return scheduler.torch_sample_scheme(logits)

        steps = []
        for st in range(4):      # the adaptive beta is state carried from one backward to the next
            scores, lens = _padded_scores(B, Nmax, gen)
            w = torch.randn(B, Nmax, 1, generator=gen) * (torch.arange(Nmax)[None, :, None] < lens[:, None, None])
            seed = 7680 + 10 * ci + st
            th = scores.clone().requires_grad_(True)
            torch.manual_seed(seed)
            res = train(th)
            (res * w).sum().backward()
            steps.append(dict(scores=scores, lens=lens, w=w, noise=_noise_from_seed(seed, (B, 1, Nmax, 1), 0.3),
                              out=res.detach().clone(), grad=th.grad.clone(), beta_after=float(target.beta),
                              grad_norm_after=float(target.grad_norm)))
        out["aimle"].append(dict(k=k, beta0=beta0, tau=tau, steps=steps))
    torch.save(out, os.path.join(OUT, "g6_sampler_grads.pt"))


def gen_mgat_train():
    """MGAT + GlobalAttention in train() mode.  The gate dropout (masking.py:159) is patched to identity for the run --
    its mask is RNG-private -- everything else is the reference's training graph incl. its custom backward rules."""
    from unittest import mock
    import torch.nn.functional as F
    from ISubGVQA.models.mgat import MGAT
    from ISubGVQA.models.att_pooling import GlobalAttention
    from ISubGVQA.models.masking import get_imle_samplers
    from ISubGVQA.sampling.methods.aimle import aimle
    from ISubGVQA.sampling.methods.imle_scheme import IMLEScheme
    from ISubGVQA.sampling.methods.noise import GumbelDistribution
    from ISubGVQA.sampling.methods.target_aimle import AdaptiveTargetDistribution

    cases = [
        dict(name="gumbel_c8", C=8, L=3, masks=[1.0, 0.15, 0.15], sampler="gumbel", k=2, kind="rand", interp=False),
        dict(name="imle_c16", C=16, L=3, masks=[0.15, 1.0, 0.15], sampler="imle", k=3, kind="rand", interp=True, beta=10.0),
        dict(name="aimle_c12", C=12, L=2, masks=[0.15, 0.15], sampler="aimle", k=2, kind="edgecases", interp=True, beta=4.0),
        dict(name="nomask_c20", C=20, L=2, masks=[1.0, 1.0], sampler="gumbel", k=2, kind="rand", interp=False),
    ]
    for ci, c in enumerate(cases):
        gen = torch.Generator().manual_seed(800 + ci)
        batch, ei = _graphs(c["kind"], gen)
        N, E, B, C, L = batch.numel(), ei.size(1), int(batch.max()) + 1, c["C"], c["L"]
        torch.manual_seed(70 + ci)
        model = MGAT(channels=C, num_ins=L, heads=4, use_instr=True, masking_thresholds=c["masks"], use_topk=True,
                     interpretable_mode=c["interp"], sampler_type=c["sampler"], sample_k=c["k"], nb_samples=1,
                     alpha=1.0, beta=c.get("beta", 10.0), tau=1.0).train()
        pool = GlobalAttention(num_node_features=C, num_out_features=C).train()
        with torch.no_grad():
            for n_, p_ in list(model.named_parameters()) + list(pool.named_parameters()):
                if n_.endswith("bias") or "bns" in n_:
                    p_.add_(0.1 * torch.randn(p_.shape, generator=gen))
        targets = {}
        for li, conv in enumerate(model.convs):
            if c["sampler"] == "imle":
                conv.mask.sampler_train, conv.mask.sampler_val = get_imle_samplers(
                    sample_k=c["k"], device="cpu", nb_samples=1, alpha=1.0, beta=c["beta"], tau=1.0)
            elif c["sampler"] == "aimle":
                scheduler = IMLEScheme("edge_candid", c["k"], 1, 1)
                targets[li] = AdaptiveTargetDistribution(initial_alpha=1.0, initial_beta=c["beta"])
                conv.mask.sampler_train = aimle(
                    scheduler.torch_sample_scheme, target_distribution=targets[li],
                    noise_distribution=GumbelDistribution(0.0, 0.3, "cpu"), nb_samples=1, theta_noise_temperature=1.0,
                    target_noise_temperature=1.0, symmetric_perturbation=True)
        x = torch.randn(N, C, generator=gen).requires_grad_(True)
        edge_attr = torch.randn(E, C, generator=gen).requires_grad_(True)
        instr = torch.randn(L, B, C, generator=gen).requires_grad_(True)
        glf = torch.randn(B, C, generator=gen).requires_grad_(True)
        w_h = torch.randn(N, C, generator=gen)
        w_e = torch.randn(B, C, generator=gen)
        seed = 9500 + ci
        torch.manual_seed(seed)
        with mock.patch.object(F, "dropout", lambda t, p=0.5, training=True, inplace=False: t):
            h, mask, _, _ = model(x=x, edge_index=ei, instr_vectors=instr, global_language_feats=glf,
                                  edge_attr=edge_attr, batch=batch, return_masks=True)
            emb, gate = pool(x=h, u=glf, batch=batch, size=None, return_mask=True, node_mask=mask)
            loss = (h * w_h).sum() + (emb * w_e).sum()
            loss.backward()
        counts = torch.bincount(batch, minlength=B)
        nmax = int(counts.max())
        torch.manual_seed(seed)
        noises = {}
        for i, thr in enumerate(c["masks"]):
            if thr != 1.0:
                noises[i] = (S.uniform_to_gumbel(torch.rand(B, nmax)) if c["sampler"] == "gumbel"
                             else S.uniform_to_gumbel(torch.rand(B, 1, nmax, 1), 0.0, 0.3))
        sd = {"gat_seq." + k: v.detach().clone() for k, v in model.state_dict().items() if "node_logits" not in k}
        sd.update({"graph_global_attention_pooling." + k: v.detach().clone() for k, v in pool.state_dict().items()})
        grads = {"gat_seq." + k: p_.grad.clone() for k, p_ in model.named_parameters() if p_.grad is not None}
        grads.update({"graph_global_attention_pooling." + k: p_.grad.clone() for k, p_ in pool.named_parameters()
                      if p_.grad is not None})
        torch.save(dict(cfg={k: c[k] for k in ("C", "L", "masks", "sampler", "k", "interp")}, beta=c.get("beta", 10.0),
                        sd=sd, x=x.detach(), edge_index=ei, edge_attr=edge_attr.detach(), batch=batch,
                        instr=instr.detach(), glf=glf.detach(), noises=noises, w_h=w_h, w_e=w_e,
                        h=h.detach(), mask=None if mask is None else mask.detach(), pool_out=emb.detach(),
                        loss=loss.detach(), grads=grads,
                        grad_x=x.grad.clone(), grad_edge_attr=edge_attr.grad.clone(), grad_instr=instr.grad.clone(),
                        grad_glf=glf.grad.clone(),
                        aimle_beta_after={li: float(t.beta) for li, t in targets.items()}),
                   os.path.join(OUT, f"g7_train_{c['name']}.pt"))


class _Data:
    """Attribute bag with the constructor of torch_geometric.data.Data (scene_graph.py:378-387 only sets attributes)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _synthetic_scene_graphs(gen):
    names = ["window", "man", "shirt", "tree", "wall", "building", "person", "sky", "table", "dog", "caf\u00e9 sign"]
    attrs = ["white", "black", "green", "large", "wooden", "small", "blue", "tall"]
    rels = ["to the left of", "to the right of", "on", "wearing", "near", "holding", "behind"]
    oov = ["zeppelin", "glorp"]

    def pick(lst):
        return lst[int(torch.randint(0, len(lst), (1,), generator=gen))]

    graphs = {}
    for g in range(14):
        n = int(torch.randint(2, 9, (1,), generator=gen))
        ids = [str(int(torch.randint(100000, 4000000, (1,), generator=gen))) for _ in range(n)]
        ids = list(dict.fromkeys(ids))
        objs = {}
        for oid in ids:
            na = int(torch.randint(0, 3, (1,), generator=gen)) if g % 3 else int(torch.randint(0, 6, (1,), generator=gen))
            a = [pick(attrs + oov[:1]) for _ in range(na)]
            if g % 3 != 0:
                a = a[:1]                       # at most one attribute: the set() order cannot matter
            rel = [{"object": pick(ids), "name": pick(rels + oov[1:])} for _ in range(int(torch.randint(0, 4, (1,), generator=gen)))]
            o = {"name": pick(names + oov), "attributes": a, "relations": rel,
                 "x": int(torch.randint(0, 500, (1,), generator=gen)), "y": 3, "w": 40, "h": 50}
            if g % 4 == 1:
                o.update(x1=int(torch.randint(0, 300, (1,), generator=gen)), y1=7, x2=311, y2=int(torch.randint(0, 300, (1,), generator=gen)))
            objs[oid] = o
        graphs[f"img{g}"] = {"width": 500, "height": 333, "objects": objs}
    graphs["empty"] = {"width": 1, "height": 1, "objects": {}}                        # -> 2-node dummy
    graphs["single"] = {"objects": {"7": {"name": "dog", "attributes": [], "relations": []}}}   # 1 edge -> 6-node dummy
    graphs["selfrel"] = {"objects": {"5": {"name": "man", "attributes": ["tall", "tall"], "relations": [
        {"object": "5", "name": "near"}, {"object": "9", "name": "on"}, {"object": "9", "name": "on"}]},
        "9": {"name": "table", "attributes": ["wooden"], "relations": [{"object": "5", "name": "near"}]},
        "10": {"name": "wall", "attributes": [], "relations": []}}}                     # "10" sorts before "5"
    return graphs, [names[:6], attrs, rels, names[3:] + ["pokemon"], rels[:2], attrs[:3]]


def gen_loader():
    import json
    import types
    tt = types.ModuleType("torchtext")
    tt_data = types.ModuleType("torchtext.data")
    tt_utils = types.ModuleType("torchtext.data.utils")
    tt_utils.get_tokenizer = lambda *a, **k: None
    tt_vocab = types.ModuleType("torchtext.vocab")
    tt_vocab.GloVe = tt_vocab.vocab = None
    sys.modules.update({"torchtext": tt, "torchtext.data": tt_data, "torchtext.data.utils": tt_utils,
                        "torchtext.vocab": tt_vocab})
    sys.modules["torch_geometric"].data.Data = _Data
    sys.modules["torch_geometric.data"].Data = _Data
    from ISubGVQA.datasets.scene_graph import GQASceneGraphs
    from . import loader as L

    gen = torch.Generator().manual_seed(4242)
    graphs, token_lists = _synthetic_scene_graphs(gen)
    stoi = L.build_vocab(token_lists)          # torchtext is absent: the vocabulary comes from the restatement

    class _V:
        def get_stoi(self):
            return stoi

    me = types.SimpleNamespace(vocab_sg=_V(), obj_mapping={}, attr_mapping={}, rel_mapping={}, scene_graphs=graphs)
    me.convert_one_gqa_scene_graph = lambda sg: GQASceneGraphs.convert_one_gqa_scene_graph(me, sg)
    per_image = {}
    for key in list(graphs) + ["not-in-the-file"]:
        d = GQASceneGraphs.query_and_translate(me, key)
        per_image[key] = dict(x=d.x.clone(), edge_index=d.edge_index.clone(), edge_attr=d.edge_attr.clone(),
                              x_bbox=d.x_bbox.clone(), added_sym_edge=d.added_sym_edge.clone())
    torch.save(dict(json=json.dumps(graphs), token_lists=token_lists, stoi=stoi, per_image=per_image),
               os.path.join(OUT, "g8_loader.pt"))


def gen_simple():
    import math
    import tempfile
    assert os.environ.get("TORCHDYNAMO_DISABLE") == "1", "run with TORCHDYNAMO_DISABLE=1 (simple.py uses torch.compile)"
    sys.modules["torch_geometric"].utils.index_sort = lambda *a, **k: None       # tensor_utils.py imports the name only
    from ISubGVQA.sampling.methods.simple_scheme import EdgeSIMPLEBatched
    cases = []
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:      # Layer() pickles its circuit into ./simple_configs (simple.py:122-130)
        os.chdir(tmp)
        try:
            for ci, (B, Nmax, k) in enumerate([(3, 6, 2), (5, 8, 5), (6, 20, 5), (4, 33, 3), (2, 4, 5), (3, 16, 1),
                                               (8, 24, 5), (1, 9, 5)]):
                gen = torch.Generator().manual_seed(900 + ci)
                scores = torch.randn(B, Nmax, 1, generator=gen)
                lens = [Nmax]
                for b in range(1, B):
                    n = int(torch.randint(1, Nmax + 1, (1,), generator=gen))
                    scores[b, n:] = 0.0
                    lens.append(n)
                w = torch.randn(B, Nmax, 1, generator=gen) * (torch.arange(Nmax)[None, :, None] < torch.tensor(lens)[:, None, None])
                sampler = EdgeSIMPLEBatched(k=k, device="cpu", policy="edge_candid")
                for train in (True, False):
                    th = scores.clone().requires_grad_(True)
                    seed = 9900 + 2 * ci + int(train)
                    torch.manual_seed(seed)
                    mask, marg = sampler(th, train=train)
                    (torch.nan_to_num(mask.squeeze(0)) * w).sum().backward()
                    n2 = 2 ** math.ceil(math.log2(Nmax))
                    torch.manual_seed(seed)
                    uniform = torch.rand(1, B, n2)
                    cases.append(dict(scores=scores, lens=torch.tensor(lens), k=k, train=train, w=w, uniform=uniform,
                                      mask=mask.detach().clone(), marginals=marg.detach().clone(), grad=th.grad.clone()))
        finally:
            os.chdir(cwd)
    torch.save(cases, os.path.join(OUT, "g9_simple.pt"))


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.dont_write_bytecode = True
    pyg_standin.install()
    sys.path.insert(0, REF)
    gen_samplers()
    gen_question()
    gen_mgat()
    gen_sampler_grads()
    gen_mgat_train()
    gen_loader()
    gen_simple()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
