"""Host-side checks that need no GPU: the C-ABI library loads and exports every declared symbol,
the modules keep the reference's state_dict layout, the product path refuses to run on the CPU."""
import argparse
import glob
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden


def test_library_exports_every_symbol_in_header():
    import __graft_entry__ as ge
    ge.build()
    from isubgvqa_amd import _lib
    header = open(os.path.join(ROOT, "include", "isg.h")).read()
    declared = set(re.findall(r"\b(isg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.isg_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define ISG_ABI_VERSION (\d+)", header).group(1))
    assert lib.isg_status_string(-2) == b"unsupported shape"
    assert lib.isg_csr_workspace_bytes(10, 7) == (2 * 11 + 7 + 2) * 4


def test_product_path_fails_loudly_on_cpu_tensors():
    from isubgvqa_amd import _lib, ops
    with pytest.raises(_lib.IsgError):
        ops.instr_gate(torch.zeros(4, 8), torch.zeros(2, 8), torch.zeros(4, dtype=torch.long))
    with pytest.raises(_lib.IsgError):
        ops.GraphPlan.build(torch.zeros(4, dtype=torch.long), num_graphs=1, max_nodes=4)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), path
        assert "/root/reference" not in src, path


G5 = sorted(glob.glob(os.path.join(GOLDEN, "g5_mgat_*.pt")))


@pytest.mark.parametrize("path", G5, ids=[os.path.basename(p) for p in G5])
def test_mgat_and_pooling_accept_reference_state_dict(path):
    from isubgvqa_amd.models import MGAT, GlobalAttention
    g = torch.load(path, map_location="cpu", weights_only=False)
    c = g["cfg"]
    m = MGAT(channels=c["C"], num_ins=c["L"], heads=4, use_instr=True, masking_thresholds=c["masks"], use_topk=True,
             interpretable_mode=c["interp"], sampler_type=c["sampler"], sample_k=c["k"])
    sd = {k[len("gat_seq."):]: v for k, v in g["sd"].items() if k.startswith("gat_seq.")}
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith("node_logits.") for k in res.missing_keys), res.missing_keys   # dropped from the fixture
    p = GlobalAttention(c["C"], c["C"])
    p.load_state_dict({k[len("graph_global_attention_pooling."):]: v for k, v in g["sd"].items()
                       if k.startswith("graph_global_attention_pooling.")}, strict=True)


def test_question_modules_accept_reference_state_dict():
    from isubgvqa_amd.models import CLIPTextEmbeddings, QuestionDecoder, QuestionEncoder
    g = load_golden("g4_question.pt")
    enc = QuestionEncoder(CLIPTextEmbeddings(50, 32, 77), 32, 32, g["nhead"], 64, 2, 0.1)
    res = enc.load_state_dict({k[len("question_encoder."):]: v for k, v in g["sd"].items()
                               if k.startswith("question_encoder.")}, strict=False)
    assert not res.unexpected_keys and res.missing_keys == ["pos_encoder.pe"], res
    dec = QuestionDecoder(4, 32, g["nhead"], 64, 2, 0.1)
    dec.load_state_dict({k[len("program_decoder."):]: v for k, v in g["sd"].items()
                         if k.startswith("program_decoder.")}, strict=True)


def _args(**kw):
    d = dict(text_sampling=False, general_hidden_dim=300, distributed=False, mgat_layers=4, use_all_instrs=False,
             use_global_mask=False, node_classification=False, sampler_type="imle", sample_k=5, nb_samples=1,
             alpha=1.0, beta=10.0, tau=1.0, use_masking=True, use_instruction=1, use_mgat=True,
             mgat_masks=[1.0, 1.0, 1.0, 0.15], use_topk=True, interpretable_mode=False, concat_instr=0, embed_cat=0,
             device="cpu", text_vocab_size=64, sg_vocab_size=40)
    d.update(kw)
    return argparse.Namespace(**d)


def test_isubgvqa_state_dict_layout_matches_appendix_c():
    from isubgvqa_amd.models import build_model
    m = build_model(_args(), None)
    sd = m.state_dict()
    C, H = 300, 4
    expect = {
        "scene_graph_encoder.sg_vocab_embedding.weight": (40, 300),
        "scene_graph_encoder.scene_graph_encoding_layer.edge_model.edge_mlp.0.weight": (C, 900),
        "scene_graph_encoder.scene_graph_encoding_layer.node_model.node_mlp_1.0.weight": (C, 600),
        "scene_graph_encoder.scene_graph_encoding_layer.node_model.node_mlp_2.2.weight": (C, C),
        "scene_graph_encoder.graph_layer_norm.mean_scale": (300,),
        "scene_graph_encoder.bbox_encoding.0.running_mean": (4,),
        "scene_graph_encoder.bbox_encoding.4.weight": (32, 16),
        "scene_graph_encoder.feat_reduc.1.weight": (300, 332),
        "text_vocab_embedding.token_embedding.weight": (64, 512),
        "question_encoder.text_vocab_embedding.position_embedding.weight": (77, 512),
        "question_encoder.emb_proj.weight": (512, 512),
        "question_encoder.pos_encoder.pe": (5000, 1, 512),
        "question_encoder.transformer_encoder.layers.3.self_attn.in_proj_weight": (1536, 512),
        "question_encoder.transformer_encoder.layers.0.linear1.weight": (2048, 512),
        "question_encoder.transformer_encoder.norm.weight": (512,),
        "program_decoder.query_embed.weight": (4, 512),
        "program_decoder.coarse_decoder.layers.2.multihead_attn.out_proj.weight": (512, 512),
        "program_decoder.coarse_decoder.layers.0.norm3.bias": (512,),
        "gat_seq.convs.3.att": (1, H, C),
        "gat_seq.convs.0.bias": (H * C,),
        "gat_seq.convs.0.lin_l.weight": (H * C, C),
        "gat_seq.convs.0.lin_r.bias": (H * C,),
        "gat_seq.convs.0.lin_edge.weight": (H * C, C),
        "gat_seq.convs.0.mask.gate_nn.2.weight": (1, C),
        "gat_seq.convs.0.mask.node_nn.0.weight": (C, C),
        "gat_seq.convs.0.mask.ques_nn.0.bias": (C,),
        "gat_seq.convs.0.mask.gate_top.select.weight": (1, C),
        "gat_seq.x_proj.1.0.weight": (H * C // 2, H * C),
        "gat_seq.x_proj.1.2.weight": (C, H * C // 2),
        "gat_seq.bns.2.mean_scale": (C,),
        "gat_seq.node_logits.2.weight": (2577, 512),
        "graph_global_attention_pooling.gate_nn.2.weight": (1, C),
        "graph_global_attention_pooling.node_nn.2.weight": (C, C),
        "graph_global_attention_pooling.ques_nn.0.weight": (C, C),
        "qsts_reduction.0.weight": (C, 2048),
        "instr_reduction.0.weight": (C, 512),
        "embedding.0.weight": (512, 3 * C),
        "logit_fc.weight": (1842, 512),
    }
    for k, shp in expect.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape))
    assert "gat_seq.convs.0.lin_edge.bias" not in sd
    # the CLIP embedding module is shared, so its tensors appear under both prefixes (isubgvqa.py:120,126-127)
    assert sd["text_vocab_embedding.token_embedding.weight"].data_ptr() == \
        sd["question_encoder.text_vocab_embedding.token_embedding.weight"].data_ptr()
    # a DDP checkpoint carries a "module." prefix (train_loop.py:89): stripping it must load strictly
    ddp = {"module." + k: v for k, v in sd.items()}
    m2 = build_model(_args(), None)
    m2.load_state_dict({k[len("module."):]: v for k, v in ddp.items()}, strict=True)


def test_forward_requires_return_masks_like_the_reference():
    from isubgvqa_amd.models import build_model
    m = build_model(_args(), None).eval()
    with pytest.raises(ValueError):
        m(None, None, None, None, None, None, return_masks=False)


def test_synthetic_cfg2_shapes():
    from isubgvqa_amd import synthetic
    cfg = synthetic.WorkloadConfig(num_graphs=512)
    wl = synthetic.make_workload(cfg)
    N, E = wl.x.size(0), wl.edge_index.size(1)
    assert 18.0 < N / 512 < 22.0 and 45.0 < E / 512 < 55.0
    assert torch.equal(wl.batch, wl.batch.sort().values)
    b = wl.batch
    assert torch.equal(b[wl.edge_index[0]], b[wl.edge_index[1]])          # edges stay inside their graph
    assert wl.edge_index.min() >= 0 and wl.edge_index.max() < N
    n = torch.bincount(b)
    assert n.min() >= 4 and n.max() <= 48 and wl.max_nodes == int(n.max())
    # every node has its self-loop
    loops = wl.edge_index[0] == wl.edge_index[1]
    assert torch.unique(wl.edge_index[0][loops]).numel() == N
    wl5 = synthetic.make_workload(synthetic.WorkloadConfig(num_graphs=256, nodes_dist="pareto", nodes_min=8,
                                                           nodes_max=200, edges_per_graph=0.0, degree="powerlaw"))
    deg = torch.bincount(wl5.edge_index[1], minlength=wl5.x.size(0))
    assert deg.max() > 8 * deg.float().mean()                            # hubs exist


def test_reference_checkpoint_reader_roundtrip(tmp_path):
    """A checkpoint in the reference's layout (train_loop.py:84-130: DDP 'module.' prefix, pickled Namespace) loads
    strictly into the drop-in model."""
    from isubgvqa_amd.checkpoint import load_model, read_checkpoint
    from isubgvqa_amd.models import build_model
    torch.manual_seed(3)
    src = build_model(_args(sampler_type="gumbel"), None)
    args = _args(sampler_type="gumbel")
    del args.nb_samples                                   # an older Namespace without this flag
    path = os.path.join(tmp_path, "checkpoint.pth")
    torch.save({"model": {"module." + k: v for k, v in src.state_dict().items()}, "optimizer": {"state": {}},
                "lr_scheduler": {}, "epoch": 7, "args": args}, path)
    sd, got_args, rest = read_checkpoint(path)
    assert rest["epoch"] == 7 and got_args.nb_samples == 1 and not any(k.startswith("module.") for k in sd)
    model, _, _ = load_model(path, device="cpu")
    assert not model.training
    for k, v in src.state_dict().items():
        assert torch.equal(model.state_dict()[k], v), k


def test_docs_quote_the_headers_symbol_count_and_abi_version():
    """README.md / DESIGN.md / INTEGRATION.md quote the number of C-ABI symbols and the ABI version; both drifted more than
    once while entry points were added.  They are checked against include/isg.h."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "isg.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    n_sym = len(set(re.findall(r"\b(isg_[a-z0-9_]+)\s*\(", body)))
    abi = int(re.search(r"#define ISG_ABI_VERSION (\d+)", hdr).group(1))
    readme = open(os.path.join(root, "README.md")).read()
    m = re.search(r"\((\d+) symbols, ABI v(\d+)\)", readme)
    assert m, "README.md no longer states '(N symbols, ABI vM)'"
    assert (int(m.group(1)), int(m.group(2))) == (n_sym, abi), f"README says {m.groups()}, header has {n_sym} symbols, ABI v{abi}"
    for name in ("DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(root, name)).read()
        for q in re.findall(r"(\d+) (?:`extern \"C\"` )?symbols", text) + re.findall(r"for all (\d+)\b", text):
            if 30 <= int(q) <= 200:      # counts of the device library (the loader's 15 are quoted too)
                assert int(q) == n_sym, f"{name} quotes {q} symbols, include/isg.h declares {n_sym}"


def test_library_holds_no_cross_selecting_packed_fp32_operation():
    """on gfx950 a v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 that takes a LOW-half operand from the HIGH dword of
    a register pair (op_sel:[..1..]) intermittently lost its low-half result in isg_gatv2_tile_conv (tools/flake/: 48-417 wrong
    launches of 1600 in every variant with the form, 0 of 1600 without).  The form is the compiler's choice, so the guard reads the
    BUILT library: every gfx950 code object is disassembled and must hold none (round 4 had 31 in five kernels)."""
    import importlib.util
    import __graft_entry__ as ge
    ge.build()
    spec = importlib.util.spec_from_file_location("scan_pk_cross", os.path.join(ROOT, "tools", "scan_pk_cross.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    # the scanner sees the form where it is (a line of round 4's failing loop) and only there
    sample = ["0000 <_Zkernel>:", "\tv_pk_fma_f32 v[34:35], v[44:45], v[52:53], v[34:35] op_sel_hi:[1,0,1] // 0001: AA",
              "\tv_pk_fma_f32 v[34:35], v[48:49], v[52:53], v[34:35] op_sel:[0,1,0] // 0002: BB",
              "\tv_pk_mul_f32 v[20:21], v[228:229], v[20:21] op_sel:[1,0]", "\tv_pk_add_f32 v[2:3], v[4:5], v[6:7]",
              "\tv_pk_fma_f16 v1, v2, v3, v4 op_sel:[0,1,0]"]
    total, hits = scan.scan_text(sample, re.compile(r"^[0-9a-f]+ <(\S+)>:"))
    assert total == 4 and [h[1].split()[0] for h in hits] == ["v_pk_fma_f32", "v_pk_mul_f32"] and hits[0][0] == "_Zkernel"
    if not os.path.exists(scan.OBJDUMP):
        pytest.skip(f"{scan.OBJDUMP} is not installed: the built libraries cannot be disassembled here")
    # both shipped libraries: tests/test_gpu_strict.py demands equal bits from the strict build, so a cross-selecting operation that
    # exists only there would show up as a "schedule bug" of the fast one
    for lib in (scan.LIB, scan.STRICT_LIB):
        objects, total, hits = scan.scan_library(lib)
        assert objects >= 18 and total > 10000, (lib, objects, total)    # the whole library was read, not an empty extraction
        assert not hits, os.path.basename(lib) + ":\n" + "\n".join(f"{scan.demangle(k)}: {t}" for k, t in hits[:10])


def test_kernel_argument_structs_are_built_by_one_complete_initialiser():
    """Round 5's GPU fault was a kernel-argument struct filled field by field with one assignment lost; the
    in-process suite passed because the stack slot still held the previous call's pointers.  Since round 6 every `*Args` struct
    that is passed to a kernel is built by ONE braced initialiser and the build refuses a field left out
    (-Werror=missing-field-initializers in HIP_FLAGS).  This test holds the three parts of that together:
      (1) the flag is in the flags every source is compiled with, and it does refuse an incomplete initialiser (hipcc, syntax only);
      (2) no source declares an argument struct without an initialiser, or with the empty one, and fills it afterwards;
      (3) every designated initialiser names every field of its struct (what the compiler enforces, read from the text), and every
          pointer the struct hands to a kernel is either null-checked on the STRUCT (`!a.field`) before the launch or documented
          as optional (`NULL` / `optional` / `may be` in the field's comment)."""
    import subprocess
    import tempfile
    import __graft_entry__ as ge
    assert "-Werror=missing-field-initializers" in ge.HIP_FLAGS
    probe = ("struct PArgs { const float *p; const int *q; int n; };\n__global__ void k(PArgs a) { if (a.q) *(float *)a.p = a.n; }\n"
             "void f(const float *p) { PArgs a = {.p = p, .n = 1}; k<<<1, 1>>>(a); }\n"
             "void g(const float *p) { PArgs a{p}; k<<<1, 1>>>(a); }\n")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "probe.hip")
        open(src, "w").write("#include <hip/hip_runtime.h>\n" + probe)
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *ge.HIP_FLAGS, "--cuda-host-only", "-fsyntax-only", src],
                           capture_output=True, text=True)
    assert r.returncode != 0 and r.stderr.count("missing field 'q' initializer") >= 2, r.stderr[-2000:]
    csrc = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
    structs, texts = {}, {}
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".hpp")):
            continue
        text = open(os.path.join(csrc, f)).read()
        texts[f] = text
        for m in re.finditer(r"struct (\w+Args) \{(.*?)\n\};", text, flags=re.S):
            fields = []          # (name, is_pointer, optional)
            for line in m.group(2).split("\n"):
                code, _, comment = line.partition("//")
                code = code.strip().rstrip(";")
                if not code:
                    if fields and comment:          # a comment line continues the previous field's description
                        fields[-1][2] = fields[-1][2] or bool(re.search(r"NULL|optional|may be", comment))
                    continue
                names = re.findall(r"(\*?)\s*(\w+)\s*(?:,|$)", code.split(None, 1)[1] if " " in code else code)
                first_ptr = "*" in code.split(",")[0]
                for i, (star, name) in enumerate(names):
                    fields.append([name, bool(star) or (i == 0 and first_ptr), bool(re.search(r"NULL|optional|may be", comment))])
            structs[m.group(1)] = fields
    assert len(structs) >= 13, sorted(structs)
    for f, text in texts.items():
        body = re.sub(r"struct \w+ \{.*?\n\};", "", text, flags=re.S)          # a member `P3Args p;` of another struct is not a fill site
        for name in structs:
            bad = re.findall(rf"\b(?:isg::)?{name} \w+(?: = \{{\}})?;", body)
            assert not bad, f"{f}: `{bad[0]}` -- build the struct with one initialiser that names every field"
        for m in re.finditer(r"\b(?:isg::)?(\w+Args) (\w+) = \{\s*\.(.*?)\};", body, flags=re.S):
            name, var, init = m.group(1), m.group(2), "." + m.group(3)
            named = re.findall(r"\.(\w+) =", init)
            want = [fl[0] for fl in structs[name]]
            assert named == want, f"{f}: {name} initialiser names {named}, the struct declares {want}"
            after = body[m.end():m.end() + 1500]
            for fname, is_ptr, optional in structs[name]:
                if is_ptr and not optional and name != "Q3Args":
                    assert re.search(rf"!{var}\.{fname}\b", after), f"{f}: {name}.{fname} is handed to a kernel without `!{var}.{fname}` behind the initialiser"


def test_switches_are_one_frozen_object_swapped_atomically():
    """ops' A/B switches live in ONE frozen dataclass (ops.CFG): the historical spelling `ops.NAME = value` (tests, tools, bench.py)
    replaces the whole object, `ops.NAME` reads the field, `with ops.configured(...)` restores; a field cannot be written in place."""
    import dataclasses
    from isubgvqa_amd import ops
    before = ops.CFG
    assert dataclasses.is_dataclass(before) and ops.SPLIT_FORWARD is before.split_forward
    with pytest.raises(dataclasses.FrozenInstanceError):
        before.split_forward = False
    try:
        ops.SPLIT_FORWARD = not before.split_forward
        assert ops.CFG is not before and ops.CFG.split_forward == (not before.split_forward) and before.split_forward != ops.SPLIT_FORWARD
        assert ops.CFG.mixed_min_nodes == before.mixed_min_nodes            # every other field carried over
        with ops.configured(mixed_min_nodes=7, gemm_kernel="panel") as cfg:
            assert ops.MIXED_MIN_NODES == 7 and ops.GEMM_KERNEL == "panel" and cfg is ops.CFG
        assert ops.MIXED_MIN_NODES == before.mixed_min_nodes and ops.GEMM_KERNEL == before.gemm_kernel
        with pytest.raises(TypeError):
            with ops.configured(no_such_switch=1):
                pass
        assert not hasattr(ops, "NO_SUCH_SWITCH")
    finally:
        ops.CFG = before
    assert ops.CFG is before
