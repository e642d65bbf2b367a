"""Training of the hot path on a real MI355X: gradients of the HIP path against
  * the CPU oracle differentiated by torch autograd (with the reference's custom backward rules restated), and
  * gradients produced by the real reference (goldens G6 samplers, G7 MGAT + pooling in train() mode).
Tolerances: masks / I-MLE gradients are integers and must be exact; fp32 gradients within 2e-4 relative to the tensor's
largest entry (summation order differs: CSR order on the GPU, edge order on the CPU)."""
import glob
import os

import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X (run with -m gpu on the GPU box)"
    return torch.device("cuda:0")


def close(got, ref, tol=2e-4, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max()) / scale
    assert err < tol, f"{what}: max |diff| / max |ref| = {err:.3e} (tol {tol})"


def random_graph_batch(sizes, extra, seed, hub=0):
    gen = torch.Generator().manual_seed(seed)
    batch, src, dst, off = [], [], [], 0
    for g, n in enumerate(sizes):
        batch += [g] * n
        for v in range(n):
            src.append(off + v); dst.append(off + v)
        for _ in range(extra * n):
            src.append(off + int(torch.randint(0, n, (1,), generator=gen)))
            dst.append(off + int(torch.randint(0, n, (1,), generator=gen)))
        if g == 0:
            for _ in range(hub):          # many edges into node 0 of graph 0: beyond the per-wave LDS strip
                src.append(off + int(torch.randint(0, n, (1,), generator=gen))); dst.append(off)
        off += n
    ei = torch.tensor([src, dst], dtype=torch.long)
    return torch.tensor(batch, dtype=torch.long), ei[:, torch.randperm(ei.size(1), generator=gen)]


@pytest.mark.parametrize("mask_kind", ["none", "node", "edge"])
@pytest.mark.parametrize("H,C,hub", [(4, 8, 0), (4, 128, 0), (4, 300, 50), (2, 16, 0), (1, 32, 40), (8, 12, 0)])
def test_gatv2_mp_backward_matches_oracle_autograd(dev, mask_kind, H, C, hub):
    from isubgvqa_amd import ops
    from oracle import model as OM
    batch, ei = random_graph_batch([5, 1, 9, 17, 3], 2, seed=H * 1000 + C, hub=hub)
    N, E = batch.numel(), ei.size(1)
    gen = torch.Generator().manual_seed(7)
    x_l, x_r = torch.randn(N, H * C, generator=gen), torch.randn(N, H * C, generator=gen)
    e_proj, att = torch.randn(E, H * C, generator=gen), torch.randn(1, H, C, generator=gen)
    bias = torch.randn(H * C, generator=gen)
    w = torch.randn(N, H * C, generator=gen)
    node_mask = (torch.rand(N, 1, generator=gen) < 0.6).float()
    edge_mask = (torch.rand(E, 1, generator=gen) < 0.6).float() * (0.5 + torch.rand(E, 1, generator=gen))

    cpu = [t.clone().requires_grad_(True) for t in (x_l, x_r, e_proj, att, bias)]
    m_cpu = None
    if mask_kind == "node":
        m_cpu = node_mask.clone().requires_grad_(True)
        em = OM.node_mask_to_edge_mask(m_cpu, ei)
    elif mask_kind == "edge":
        m_cpu = edge_mask.clone().requires_grad_(True)
        em = m_cpu
    else:
        em = None
    out_ref, _ = OM.gatv2_message_passing(cpu[0].view(N, H, C), cpu[1].view(N, H, C), cpu[2].view(E, H, C), cpu[3], ei, em)
    ((out_ref.reshape(N, H * C) + cpu[4]) * w).sum().backward()

    gpu = [t.to(dev).requires_grad_(True) for t in (x_l, x_r, e_proj, att, bias)]
    plan = ops.GraphPlan.build(batch.to(dev), ei.to(dev), num_graphs=5)
    m_gpu = None
    kw = {}
    if mask_kind == "node":
        m_gpu = node_mask.to(dev).requires_grad_(True)
        kw["node_mask"] = m_gpu
    elif mask_kind == "edge":
        m_gpu = edge_mask.to(dev).requires_grad_(True)
        kw["edge_mask"] = m_gpu
    out, alpha = ops.gatv2_mp(gpu[0], gpu[1], gpu[2], gpu[3], plan, H, bias=gpu[4], **kw)
    assert not alpha.requires_grad
    close(out, out_ref.reshape(N, H * C) + cpu[4], 1e-5, "forward")
    (out * w.to(dev)).sum().backward()
    for name, a, b in zip(("d x_l", "d x_r", "d e_proj", "d att", "d bias"), gpu, cpu):
        close(a.grad, b.grad, 2e-4, name)
    if m_gpu is not None:
        close(m_gpu.grad, m_cpu.grad, 2e-4, f"d {mask_kind} mask")


def test_node_to_edge_mask_backward_keeps_the_reference_rule(dev):
    from isubgvqa_amd.sampling.node_edge_masks import NodeMaskToEdgeMask
    _, ei = random_graph_batch([6, 4, 11], 3, seed=3)
    N, E = 21, ei.size(1)
    gen = torch.Generator().manual_seed(5)
    mask = torch.rand(N, 1, generator=gen)
    w = torch.randn(E, 1, generator=gen)
    m = mask.to(dev).requires_grad_(True)
    em = NodeMaskToEdgeMask.apply(m, ei.to(dev), torch.tensor(N))
    assert torch.equal(em.cpu(), mask[ei[0]] * mask[ei[1]])
    (em * w.to(dev)).sum().backward()
    ref = torch.zeros(N, 1).index_add_(0, ei[1], w)          # destination only, no product rule
    close(m.grad, ref, 1e-6, "d mask")


# ---- samplers against gradients produced by the reference (G6) -------------------------------------------------------
def _ragged(c, dev):
    """(scores[N,1], batch[N], plan, slot index of every real node in the dense [B*Nmax] layout)"""
    from isubgvqa_amd import ops
    lens = c["lens"]
    B, nmax = c["scores"].shape[:2]
    batch = torch.repeat_interleave(torch.arange(B), lens)
    pos = torch.cat([torch.arange(int(n)) for n in lens])
    slot = batch * nmax + pos
    plan = ops.GraphPlan.build(batch.to(dev), None, num_graphs=B)
    return slot, batch, plan


def test_gumbel_straight_through_gradient_matches_reference(dev):
    from isubgvqa_amd import ops
    from isubgvqa_amd.sampling.methods.gumbel_scheme import GumbelSampler
    for c in load_golden("g6_sampler_grads.pt")["gumbel"]:
        B, nmax = c["scores"].shape[:2]
        # dense, through the drop-in sampler class
        th = c["scores"].to(dev).requires_grad_(True)
        res, _ = GumbelSampler(k=c["k"], policy="edge_candid", train_ensemble=1, val_ensemble=1)(
            th, train=True, noise=c["noise"].to(dev))
        assert torch.equal(res.detach().cpu() > 0.5, c["out"] > 0.5)
        (res.squeeze(0) * c["w"].to(dev)).sum().backward()
        close(th.grad, c["grad"], 2e-4, "gumbel dense grad")
        # ragged rows (what MaskingModel runs): rows of different lengths inside one padded batch
        if int(c["lens"].max()) == nmax:
            slot, _, plan = _ragged(c, dev)
            sc = c["scores"].reshape(-1)[slot].view(-1, 1).to(dev).requires_grad_(True)
            out = ops.topk_gumbel(sc, c["k"], 0.1, plan=plan, noise=c["noise"].to(dev))
            assert torch.equal(out.detach().cpu().view(-1) > 0.5, c["out"].reshape(-1)[slot] > 0.5)
            (out * c["w"].reshape(-1)[slot].view(-1, 1).to(dev)).sum().backward()
            close(sc.grad.view(-1), c["grad"].reshape(-1)[slot], 2e-4, "gumbel ragged grad")


def test_imle_second_map_solve_matches_reference(dev):
    from isubgvqa_amd.models.masking import get_imle_samplers
    for c in load_golden("g6_sampler_grads.pt")["imle"]:
        B, nmax = c["scores"].shape[:2]
        train, _ = get_imle_samplers(sample_k=c["k"], device=dev, nb_samples=1, alpha=1.0, beta=c["beta"], tau=1.0)
        th = c["scores"].to(dev).requires_grad_(True)
        res = train(th, noise=c["noise"].to(dev))[0]
        assert torch.equal(res.detach().cpu(), c["out"])
        (res.squeeze(0) * c["w"].to(dev)).sum().backward()
        assert torch.equal(th.grad.cpu(), c["grad"]), "I-MLE gradient (z - z') differs from the reference"
        if int(c["lens"].max()) == nmax:
            slot, _, plan = _ragged(c, dev)
            sc = c["scores"].reshape(-1)[slot].view(-1, 1).to(dev).requires_grad_(True)
            out = train.differentiable(sc, plan, c["noise"].reshape(B, nmax).to(dev))
            (out * c["w"].reshape(-1)[slot].view(-1, 1).to(dev)).sum().backward()
            assert torch.equal(sc.grad.cpu().view(-1), c["grad"].reshape(-1)[slot])


def test_aimle_adaptive_target_matches_reference_over_steps(dev):
    from isubgvqa_amd.sampling.methods.aimle import aimle
    from isubgvqa_amd.sampling.methods.deterministic_scheme import IMLEScheme
    from isubgvqa_amd.sampling.methods.noise import GumbelDistribution
    from isubgvqa_amd.sampling.methods.target_aimle import AdaptiveTargetDistribution
    from isubgvqa_amd.models.masking import _scheme_fn
    for c in load_golden("g6_sampler_grads.pt")["aimle"]:
        target = AdaptiveTargetDistribution(initial_alpha=1.0, initial_beta=c["beta0"])
        train = aimle(_scheme_fn(IMLEScheme("edge_candid", c["k"], 1, 1)), target_distribution=target,
                      noise_distribution=GumbelDistribution(0.0, 0.3, dev), nb_samples=1,
                      theta_noise_temperature=c["tau"], target_noise_temperature=c["tau"], symmetric_perturbation=True)
        for st in c["steps"]:
            th = st["scores"].to(dev).requires_grad_(True)
            res = train(th, noise=st["noise"].to(dev))
            assert torch.equal(res.detach().cpu(), st["out"])
            (res * st["w"].to(dev)).sum().backward()
            close(th.grad, st["grad"], 1e-5, "aimle grad") if float(st["grad"].abs().max()) > 0 else \
                (lambda: None)()
            assert float(th.grad.abs().max()) > 0 or float(st["grad"].abs().max()) == 0
            assert abs(target.beta - st["beta_after"]) < 1e-9
            assert abs(target.grad_norm - st["grad_norm_after"]) < 1e-5


# ---- the MGAT stack in train() mode against the reference's gradients (G7) ------------------------------------------
G7 = sorted(glob.glob(os.path.join(GOLDEN, "g7_train_*.pt")))


def _load_g7(path, dev):
    from isubgvqa_amd.models import MGAT, GlobalAttention
    from isubgvqa_amd.sampling.methods.target_aimle import AdaptiveTargetDistribution
    g = torch.load(path, map_location="cpu", weights_only=False)
    c = g["cfg"]
    m = MGAT(channels=c["C"], num_ins=c["L"], heads=4, use_instr=True, masking_thresholds=c["masks"], use_topk=True,
             interpretable_mode=c["interp"], sampler_type=c["sampler"], sample_k=c["k"], beta=g["beta"])
    m.load_state_dict({k[len("gat_seq."):]: v for k, v in g["sd"].items() if k.startswith("gat_seq.")}, strict=False)
    p = GlobalAttention(c["C"], c["C"])
    p.load_state_dict({k[len("graph_global_attention_pooling."):]: v for k, v in g["sd"].items()
                       if k.startswith("graph_global_attention_pooling.")})
    for conv in m.convs:
        conv.mask.gate_dropout = 0.0          # the goldens were made with the gate dropout patched to identity
        if c["sampler"] == "aimle":
            conv.mask.sampler_train.target_distribution = AdaptiveTargetDistribution(initial_alpha=1.0,
                                                                                     initial_beta=g["beta"])
    return g, m.to(dev).train(), p.to(dev).train()


@pytest.mark.parametrize("path", G7, ids=[os.path.basename(p)[9:-3] for p in G7])
def test_training_step_matches_reference_gradients(dev, path):
    g, m, p = _load_g7(path, dev)
    ins = {k: g[k].to(dev).requires_grad_(True) for k in ("x", "edge_attr", "instr", "glf")}
    noises = {i: n.to(dev) for i, n in g["noises"].items()}
    h, mask, _, _ = m(x=ins["x"], edge_index=g["edge_index"].to(dev), instr_vectors=ins["instr"],
                      global_language_feats=ins["glf"], edge_attr=ins["edge_attr"], batch=g["batch"].to(dev),
                      return_masks=True, noises=noises)
    emb, _ = p(x=h, u=ins["glf"], batch=g["batch"].to(dev), size=None, return_mask=True, node_mask=mask)
    if g["mask"] is not None:
        assert torch.equal(mask.detach().cpu() > 0.5, g["mask"] > 0.5)
    close(h, g["h"], 1e-5, "h")
    loss = (h * g["w_h"].to(dev)).sum() + (emb * g["w_e"].to(dev)).sum()
    close(loss, g["loss"], 1e-4, "loss")
    loss.backward()
    got = {"gat_seq." + k: v.grad for k, v in m.named_parameters() if v.grad is not None}
    got.update({"graph_global_attention_pooling." + k: v.grad for k, v in p.named_parameters() if v.grad is not None})
    missing = [k for k, ref in g["grads"].items() if k not in got and float(ref.abs().max()) > 0]
    assert not missing, f"parameters without a gradient: {missing}"
    for k, ref in g["grads"].items():
        if k in got:
            close(got[k].reshape(ref.shape), ref, 5e-4, k)
    for k in ("x", "edge_attr", "instr", "glf"):
        close(ins[k].grad, g["grad_" + k], 5e-4, "d " + k)
    for li, beta in g["aimle_beta_after"].items():
        assert abs(m.convs[li].mask.sampler_train.target_distribution.beta - beta) < 1e-9


def test_training_step_cfg2_shape_matches_oracle(dev):
    """A cfg2-shaped batch (C=128, H=4, L=3, Gumbel k=5) small enough for CPU autograd: every parameter gradient of
    MGAT + pooling + classifier against the oracle."""
    from isubgvqa_amd import synthetic
    from oracle import model as OM
    from test_gpu_models import _noises, _oracle_cfg
    cfg = synthetic.WorkloadConfig(num_graphs=48, seed=99, masks=(1.0, 0.15, 0.15))
    wl = synthetic.make_workload(cfg)
    model = synthetic.build_answer_model(cfg).to(dev).train()
    for conv in model.gat_seq.convs:
        conv.mask.gate_dropout = 0.0
    model.embedding[2].p = 0.0
    noises = _noises(cfg, wl, 5)
    gen = torch.Generator().manual_seed(1)
    w = torch.randn(cfg.num_graphs, 1842, generator=gen)
    logits, mask, _ = model(wl.to(dev), noises={i: n.to(dev) for i, n in noises.items()})
    (logits * w.to(dev)).sum().backward()

    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()
          if v.is_floating_point()}
    ocfg = _oracle_cfg(cfg)
    ocfg.training = True
    ref_logits, ref_mask = OM.mgat_pool_classify(sd, wl.x, wl.edge_index, wl.edge_attr, wl.batch, wl.instr, wl.glf,
                                                 ocfg, noises)[:2]
    (ref_logits * w).sum().backward()
    assert int((mask.detach().cpu().view(-1) != ref_mask.detach().view(-1)).sum()) == 0
    close(logits, ref_logits, 1e-4, "logits")
    for k, v in model.named_parameters():
        if sd[k].grad is None:
            continue
        assert v.grad is not None, k
        close(v.grad, sd[k].grad.reshape(v.grad.shape), 1e-3, k)


# ---- SIMPLE sampler against the reference (G9) ------------------------------------------------------------------------
def test_simple_sampler_matches_reference(dev):
    from isubgvqa_amd import ops
    from isubgvqa_amd.sampling.methods.simple_scheme import EdgeSIMPLEBatched
    for c in load_golden("g9_simple.pt"):
        B, nmax = c["scores"].shape[:2]
        n = c["uniform"].shape[-1]
        sampler = EdgeSIMPLEBatched(k=c["k"], device=dev, policy="edge_candid")
        th = c["scores"].to(dev).requires_grad_(True)
        mask, marg = sampler(th, train=c["train"], uniform=c["uniform"].view(B, n).to(dev))
        assert torch.equal(torch.isnan(marg.cpu()), torch.isnan(c["marginals"]))
        torch.testing.assert_close(marg.detach().cpu(), c["marginals"], rtol=2e-5, atol=2e-6, equal_nan=True)
        torch.testing.assert_close(mask.detach().cpu(), c["mask"], rtol=0, atol=2e-6, equal_nan=True)
        (torch.nan_to_num(mask.squeeze(0)) * c["w"].to(dev)).sum().backward()
        torch.testing.assert_close(th.grad.cpu(), c["grad"], rtol=2e-4, atol=5e-6, equal_nan=True)
        # ragged rows, as MaskingModel calls it (only when the padded batch really has a full-length row)
        if int(c["lens"].max()) == nmax and not torch.isnan(c["marginals"]).any():
            slot, _, plan = _ragged(c, dev)
            sc = c["scores"].reshape(-1)[slot].view(-1, 1).to(dev).requires_grad_(True)
            out = ops.simple_topk(sc, c["k"], plan=plan, uniform=c["uniform"].view(B, n).to(dev))
            torch.testing.assert_close(out.detach().cpu().view(-1), c["mask"].reshape(-1)[slot], rtol=0, atol=2e-6)
            (out * c["w"].reshape(-1)[slot].view(-1, 1).to(dev)).sum().backward()
            torch.testing.assert_close(sc.grad.cpu().view(-1), c["grad"].reshape(-1)[slot], rtol=2e-4, atol=5e-6)


def test_full_model_training_step_matches_oracle(dev):
    """The whole ISubGVQA model in train() mode (question encoder/decoder, scene-graph encoder with batch-statistics
    BatchNorm and fp64 GraphNorm, MGAT with the I-MLE estimator, pooling, classifier): every parameter gradient against
    oracle autograd.  All dropouts are set to 0 (their masks are RNG-private)."""
    import argparse
    from isubgvqa_amd import synthetic
    from isubgvqa_amd.models import build_model
    from oracle import model as OM
    from test_gpu_models import _full_args
    torch.manual_seed(0)
    args = _full_args(sampler_type="imle", mgat_masks=[1.0, 0.15, 1.0, 0.15])
    model = build_model(args, None).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if hasattr(m, "gate_dropout"):
            m.gate_dropout = 0.0
    gen = torch.Generator().manual_seed(29)
    cfg = synthetic.WorkloadConfig(num_graphs=10, nodes_dist="uniform", nodes_min=2, nodes_max=16, edges_per_graph=0.0,
                                   seed=97)
    batch, ei, nmax = synthetic.make_topology(cfg, gen)
    N, E, B, T = batch.numel(), ei.size(1), 10, 9
    x = torch.randint(0, 2578, (N, 4), generator=gen)
    edge_attr = torch.randint(0, 2578, (E,), generator=gen)
    x_bbox = torch.randint(0, 640, (N, 4), generator=gen)
    sym = torch.randint(0, 10, (12,), generator=gen)
    q = torch.randint(0, 512, (B, T), generator=gen)
    qmask = (torch.arange(T)[None] < torch.randint(5, T + 1, (B,), generator=gen)[:, None]).long()
    w = torch.randn(B, 1842, generator=gen)
    from oracle import samplers as OS
    noises = {i: OS.uniform_to_gumbel(torch.rand(B, 1, nmax, 1, generator=gen), 0.0, 0.3) for i in (1, 3)}

    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.detach().clone())
          for k, v in model.state_dict().items()}
    ocfg = OM.PathConfig(heads=4, masking_thresholds=[1.0, 0.15, 1.0, 0.15], sampler_type="imle", sample_k=5,
                         training=True, imle_beta=10.0)
    rl, rm, _, _, _ = OM.isubgvqa_forward(sd, x, ei, edge_attr, batch, q, qmask, x_bbox, sym, ocfg, noises)
    (rl * w).sum().backward()

    model = model.to(dev)
    sgd = argparse.Namespace(x_bbox=x_bbox.to(dev), added_sym_edge=sym.to(dev))
    gl, gm, _, _, _ = model(x.to(dev), ei.to(dev), edge_attr.to(dev), batch.to(dev), q.to(dev), qmask.to(dev),
                            return_masks=True, scene_graphs=sgd, noises={i: n.to(dev) for i, n in noises.items()})
    assert torch.equal(gm.detach().cpu() > 0.5, rm.detach() > 0.5)
    close(gl, rl, 1e-4, "logits")
    (gl * w.to(dev)).sum().backward()
    checked = 0
    for k, v in model.named_parameters():
        ref = sd[k].grad
        # (a bias in front of a GraphNorm has an analytically ~zero gradient: pure cancellation noise, 1e-7)
        if ref is None or float(ref.abs().max()) < 1e-5:
            continue
        assert v.grad is not None, k
        close(v.grad, ref.reshape(v.grad.shape), 2e-3, k)
        checked += 1
    assert checked > 100


# ---- per-graph backward kernels against torch autograd of the same function on the device ---------------------------
@pytest.mark.parametrize("C", [8, 64, 128, 300])
@pytest.mark.parametrize("masked", [False, True])
def test_tail_pool_and_gate_backward_kernels(dev, C, masked):
    from isubgvqa_amd import autograd as AG
    from isubgvqa_amd import ops
    sizes = [5, 1, 70, 17, 3, 0, 9]          # incl. an empty graph and one longer than a wave
    B = len(sizes)
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes)).to(dev)
    N = batch.numel()
    plan = ops.GraphPlan.build(batch, None, num_graphs=B)
    gen = torch.Generator().manual_seed(C + int(masked))
    mk = lambda *s: torch.randn(*s, generator=gen).to(dev).requires_grad_(True)
    mask = (torch.rand(N, 1, generator=gen) > 0.3).float().to(dev).requires_grad_(True) if masked else None

    def compare(run_hip, run_ref, tensors, what):
        outs = run_hip()
        outs = outs if isinstance(outs, tuple) else (outs,)
        ws = [torch.randn(o.shape, generator=gen).to(dev) for o in outs]
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        got = [t.grad.clone() for t in tensors]
        for t in tensors:
            t.grad = None
        refs = run_ref()
        refs = refs if isinstance(refs, tuple) else (refs,)
        for o, r in zip(outs, refs):
            close(o, r, 2e-5, what + " forward")
        sum((o * w).sum() for o, w in zip(refs, ws)).backward()
        for i, (t, gt) in enumerate(zip(tensors, got)):
            close(gt, t.grad, 3e-4, f"{what} grad #{i}")
            t.grad = None

    # layer tail
    ins, c, h = mk(B, C), mk(N, C), mk(N, C)
    w, b, ms = mk(C), mk(C), mk(C)
    ts = [ins, c, h, w, b, ms] + ([mask] if masked else [])
    compare(lambda: ops.mgat_layer_tail(ins, c, h, plan, w, b, ms, 1e-5, node_mask=mask),
            lambda: AG._layer_tail_t(ins, c, h, w, b, ms, mask, batch, B, 1e-5), ts, "tail")
    # pooling (both outputs carry gradient)
    xn, q = mk(N, C), mk(B, C)
    ts = [xn, q] + ([mask] if masked else [])
    compare(lambda: ops.global_attn_pool(xn, q, plan, mask), lambda: AG._pool_t(xn, q, mask, batch, B), ts, "pool")
    # instruction gate
    x, instr = mk(N, C), mk(B, C)
    compare(lambda: ops.instr_gate(x, instr, batch, plan=plan), lambda: AG._instr_gate_t(x, instr, batch), [x, instr], "instr gate")
    # node gate, with the double index (several graphs share a row of q) and without
    xg, qg = mk(N, C), mk(B, C)
    for dbl in (True, False):
        compare(lambda: ops.node_gate(xg, qg, batch, dbl, plan=plan), lambda: AG._node_gate_t(xg, qg, batch, dbl),
                [xg, qg], f"node gate dbl={dbl}")


def test_simple_sampler_long_rows_against_oracle(dev):
    """Rows beyond the 64 KB default LDS window (n = 1024: 98 KB of tree per row) and k at the table limit."""
    from isubgvqa_amd.sampling.methods.simple_scheme import EdgeSIMPLEBatched
    from oracle import simple as OSI
    for (B, nmax, k) in [(3, 700, 5), (2, 130, 16), (4, 257, 1)]:
        gen = torch.Generator().manual_seed(nmax + k)
        scores = torch.randn(B, nmax, 1, generator=gen)
        scores[1, nmax // 2:] = 0.0                       # a ragged row: zero pads forced into the subset
        n = 1 << (nmax - 1).bit_length()
        uni = torch.rand(1, B, n, generator=gen)
        ref_mask, ref_marg = OSI.simple_forward(scores, k, uni)
        mask, marg = EdgeSIMPLEBatched(k=k, device=dev, policy="edge_candid")(scores.to(dev), train=False,
                                                                              uniform=uni.view(B, n).to(dev))
        assert torch.equal(torch.isnan(marg.cpu()), torch.isnan(ref_marg))
        torch.testing.assert_close(marg.cpu(), ref_marg, rtol=5e-5, atol=5e-6, equal_nan=True)
        torch.testing.assert_close(mask.cpu(), ref_mask, rtol=0, atol=5e-6, equal_nan=True)


@pytest.mark.parametrize("sampler", ["gumbel", "imle", "aimle", "simple"])
def test_a_few_optimizer_steps_reduce_the_loss(dev, sampler):
    """End-to-end training sanity on the HIP path: Adam on MGAT + pooling + classifier fits a small fixed batch."""
    from isubgvqa_amd import synthetic
    cfg = synthetic.WorkloadConfig(num_graphs=64, channels=64, layers=3, masks=(1.0, 0.15, 0.15), sampler=sampler,
                                   sample_k=5, seed=123)
    wl = synthetic.make_workload(cfg).to(dev)
    torch.manual_seed(0)
    model = synthetic.build_answer_model(cfg).to(dev).train()
    target = torch.randint(0, 1842, (cfg.num_graphs,), generator=torch.Generator().manual_seed(1)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for step in range(25):
        opt.zero_grad(set_to_none=True)
        logits, _, _ = model(wl, seed=7 + step)
        loss = torch.nn.functional.cross_entropy(logits, target)
        assert torch.isfinite(loss)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.6 * losses[0], losses[::6]


def test_mp_backward_full_size_properties(dev):
    """BASELINE configs[1] size (4096 graphs, 82 k nodes, 205 k edges, H*C = 512), where CPU autograd is too slow to be the
    checker: the backward is linear in grad_out, bitwise reproducible (no atomics), and consistent with a directional
    finite difference of the forward."""
    from isubgvqa_amd import ops, synthetic
    cfg = synthetic.CFG2
    wl = synthetic.make_workload(cfg).to(dev)
    N, E, H, C = wl.x.size(0), wl.edge_index.size(1), cfg.heads, cfg.channels
    plan = ops.GraphPlan.build(wl.batch, wl.edge_index, num_graphs=cfg.num_graphs, max_nodes=wl.max_nodes,
                               max_edges=wl.max_edges)
    g = torch.Generator(device=dev).manual_seed(3)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    x_l, x_r, e_proj, att = r(N, H * C), r(N, H * C), r(E, H * C), 0.3 * r(1, H, C)
    mask = (torch.rand(N, 1, device=dev, generator=g) > 0.3).float()
    _, alpha = ops.gatv2_mp(x_l, x_r, e_proj, att, plan, H, node_mask=mask)
    g1, g2 = r(N, H * C), r(N, H * C)
    bw = lambda go: ops.gatv2_mp_backward(x_l, x_r, e_proj, att, alpha, go, plan, H, node_mask=mask, want_mask_grad=True)
    a, b, ab, a2 = bw(g1), bw(g2), bw(g1 + g2), bw(g1)
    for i, name in enumerate(("d x_l", "d x_r", "d e_proj", "d att", "d bias", "d edge mask")):
        assert torch.equal(a[i], a2[i]), f"{name}: not bitwise reproducible"
        close(ab[i], a[i] + b[i], 2e-4, f"{name}: linearity")
    # the same function written with torch ops ON THE DEVICE (gather / segment softmax / index_add), differentiated by
    # torch autograd: an independent full-size check of every input gradient
    src, dst = wl.edge_index[0], wl.edge_index[1]
    leaves = [t.clone().requires_grad_(True) for t in (x_l, x_r, e_proj, att)]
    m_e = (mask[src] * mask[dst])
    sfeat = (leaves[1][dst] + leaves[0][src]) + leaves[2]
    sfeat = torch.nn.functional.leaky_relu(sfeat * m_e, 0.2) * m_e
    logit = (sfeat.view(E, H, C) * leaves[3]).sum(-1)
    mx = torch.full((N, H), float("-inf"), device=dev).scatter_reduce(0, dst[:, None].expand(E, H), logit.detach(), "amax")
    ex = (logit - mx[dst]).exp()
    al = ex / (torch.zeros(N, H, device=dev).index_add_(0, dst, ex)[dst] + 1e-16)
    msg = leaves[0][src].view(E, H, C) * (al * m_e).unsqueeze(-1)
    out_t = torch.zeros(N, H, C, device=dev).index_add_(0, dst, msg).view(N, H * C)
    (out_t * g1).sum().backward()
    for i, name in enumerate(("d x_l", "d x_r", "d e_proj", "d att")):
        close(a[i].view_as(leaves[i].grad), leaves[i].grad, 5e-4, f"{name} vs device autograd")
