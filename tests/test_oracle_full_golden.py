"""G10: rows A1 (ISubGVQA.forward) and A4 (SceneGraphEncoder.forward + EdgeModel / NodeModel) of SURVEY §8 pinned to the
REFERENCE ITSELF.  tests/golden/g10_full.pt was written by oracle/make_goldens.py::gen_full, which calls the reference's
unbound `ISubGVQA.forward` (models/isubgvqa.py:213-297) and `SceneGraphEncoder.forward`
(models/scene_graph_encoder.py:53-143) over the reference's own QuestionEncoder / QuestionDecoder / MGAT /
GlobalAttention at the default architecture (C = 300, d = 512).  Weights are not stored: both sides fill the state_dict
with the seeded recipe of oracle/recipe.py, and the stored checksums detect a drifted RNG stream.

What this pins against the reference (and no longer against a reading of it): the `.view` scramble of the decoder output
(quirk Q4, B > 1 so batch items mix), the un-offset `added_sym_edge` sign flip with duplicate indices (Q6), eval-mode
BatchNorm with non-trivial running statistics, MetaLayer order, scatter_mean by destination, the float64 GraphNorm (Q10).
"""
import os

import pytest
import torch

from oracle import model as OM
from oracle import recipe as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g10_full.pt")
CASES = torch.load(GOLD)


def recipe_state_dict(case):
    sd = {k: R.recipe_tensor(k, torch.empty(shape), case["seed"]) for k, shape in case["keys"].items()
          if "num_batches_tracked" not in k}
    for k, want in case["checksums"].items():
        got = float(sd[k].double().sum())
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), f"recipe stream drifted at {k}: {got} vs {want}"
    return sd


def path_config(case):
    c = case["cfg"]
    return OM.PathConfig(heads=4, masking_thresholds=list(c["masks"]), use_topk=True, sampler_type=c["sampler"],
                         sample_k=c["k"], interpretable_mode=c["interp"])


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_scene_graph_encoder_restatement_matches_the_reference(ci):
    case = CASES[ci]
    sd = recipe_state_dict(case)
    with torch.no_grad():
        x_enc, e_enc = OM.scene_graph_encoder_forward(sd, "scene_graph_encoder", case["x"], case["edge_index"],
                                                      case["edge_attr"], case["batch"], case["x_bbox"],
                                                      case["added_sym_edge"], path_config(case))
    assert torch.allclose(e_enc, case["e_enc"], atol=2e-6, rtol=1e-6)
    assert torch.allclose(x_enc, case["x_enc"], atol=5e-6, rtol=1e-6)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_full_forward_restatement_matches_the_reference(ci):
    case = CASES[ci]
    sd = recipe_state_dict(case)
    with torch.no_grad():
        enc = OM.question_encoder_forward(sd, "question_encoder", case["questions"], case["att_mask"], 8)
        dec = OM.question_decoder_forward(sd, "program_decoder", enc, 8)
        logits, mask, gate, nl, mt = OM.isubgvqa_forward(sd, case["x"], case["edge_index"], case["edge_attr"],
                                                         case["batch"], case["questions"], case["att_mask"],
                                                         case["x_bbox"], case["added_sym_edge"], path_config(case),
                                                         case["noises"] or None)
    assert torch.allclose(enc, case["enc_out"], atol=2e-5)
    assert torch.allclose(dec, case["dec_out"], atol=2e-5)
    assert nl == [] and mt is None
    assert torch.equal(mask > 0.5, case["mask"] > 0.5)                    # top-k node mask: exact
    assert torch.allclose(mask, case["mask"], atol=3e-7)                  # straight-through values
    err = (logits - case["logits"]).abs().max().item()
    print(f"G10[{ci}] oracle vs reference: max |logit diff| = {err:.3e}")
    # the text side differs from torch's fused attention (SDPA on CPU: blocked online softmax) by ~3e-6, and four
    # GraphNorm'd layers at C = 300 amplify that ~7x: half of north_star's 1e-4 is the bound end to end ...
    assert err < 5e-5
    assert torch.allclose(gate, case["gate"], atol=5e-6)
    # ... and everything AFTER the decoder (the .view scramble, reductions, scene-graph encoder, MGAT, pooling,
    # classifier) is pinned tightly by starting from the reference's own decoder output
    with torch.no_grad():
        glf, instr = OM.language_features(sd, case["dec_out"])
        x_enc, e_enc = OM.scene_graph_encoder_forward(sd, "scene_graph_encoder", case["x"], case["edge_index"],
                                                      case["edge_attr"], case["batch"], case["x_bbox"],
                                                      case["added_sym_edge"], path_config(case))
        l2, m2, g2 = OM.mgat_pool_classify(sd, x_enc, case["edge_index"], e_enc, case["batch"], instr, glf,
                                           path_config(case), case["noises"] or None)
    err2 = (l2 - case["logits"]).abs().max().item()
    print(f"G10[{ci}] downstream of the reference's decoder output: max |logit diff| = {err2:.3e}")
    assert err2 < 4e-6 and torch.equal(m2 > 0.5, case["mask"] > 0.5) and torch.allclose(g2, case["gate"], atol=1e-6)


def test_the_golden_exercises_the_quirks_it_claims():
    for case in CASES:
        assert int(case["batch"].max()) + 1 > 1                           # Q4 mixes batch items only when B > 1
        sym = case["added_sym_edge"]
        assert sym.unique().numel() < sym.numel()                         # duplicates in the sign-flip index (Q6)
        assert case["att_mask"].sum() < case["att_mask"].numel()          # ragged questions
        assert 0 < case["mask"].sum() < case["mask"].numel()              # a real sub-graph was selected
