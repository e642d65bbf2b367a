"""The C++ scene-graph loader against the reference-made golden G8 and the oracle.  No GPU."""
import json
import os
import random

import pytest
import torch

from conftest import load_golden
from oracle import loader as OL
from test_oracle_loader import same_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge
    ge.build()
    from isubgvqa_amd import loader
    return loader


@pytest.fixture(scope="module")
def g8():
    return load_golden("g8_loader.pt")


@pytest.fixture(scope="module")
def store(L, g8):
    return L.SceneGraphStore(L.SceneGraphVocab(g8["token_lists"])).add_json(g8["json"])


def test_library_exports_every_symbol_in_header(L):
    lib = L.load()
    declared = L.declared_symbols(os.path.join(ROOT, "include", "isg_loader.h"))
    assert declared == sorted(L.SIGNATURES), (declared, sorted(L.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name)
    assert lib.isg_loader_abi_version() == L.ABI_VERSION


def test_vocab_matches_reference_construction(L, g8):
    v = L.SceneGraphVocab(g8["token_lists"])
    assert v.get_stoi() == g8["stoi"]
    assert len(v) == len(g8["stoi"])
    assert v.lookup("definitely-not-a-token") == -1
    first = g8["token_lists"][0][0]
    assert v.lookup(first) == g8["stoi"].get(first, -1)          # the position-0 token is dropped (quirk)


def test_every_image_matches_the_reference_conversion(L, g8, store):
    graphs = json.loads(g8["json"])
    assert len(store) == len(graphs)
    for key, ref in g8["per_image"].items():
        b = store.collate([key], pin_memory=False)
        got = dict(x=b.x, edge_index=b.edge_index, edge_attr=b.edge_attr, x_bbox=b.x_bbox, added_sym_edge=b.added_sym_edge)
        same_graph(got, ref, graphs.get(key), g8["stoi"])
        assert b.max_nodes == ref["x"].size(0) and b.max_edges == ref["edge_index"].size(1)
        assert (key in store) == (key in graphs)


def test_collate_matches_oracle_batch(L, g8, store):
    graphs = json.loads(g8["json"])
    keys = ["img3", "selfrel", "empty", "img0", "not-in-the-file", "img3", "single", "img12"]
    ref = OL.collate([OL.dataset_item(OL.query_and_translate(graphs, k, g8["stoi"])) for k in keys])
    b = store.collate(keys, pin_memory=False)
    for name in ("x", "edge_index", "edge_attr", "x_bbox", "added_sym_edge", "batch", "ptr"):
        assert torch.equal(getattr(b, name), ref[name]), name
    assert b.num_graphs == len(keys)
    counts = torch.bincount(ref["batch"])
    assert b.max_nodes == int(counts.max())
    assert b.max_edges == int(torch.bincount(ref["batch"][ref["edge_index"][1]]).max())
    empty = store.collate([], pin_memory=False)
    assert empty.x.shape == (0, 4) and empty.ptr.tolist() == [0] and empty.num_graphs == 0


def test_mirror_classes_keep_the_reference_surface(L, g8, tmp_path):
    from isubgvqa_amd.datasets import GQASceneGraphs, gqa_collate
    graphs = json.loads(g8["json"])
    half = dict(list(graphs.items())[:9])
    override = {"img0": graphs["img5"]}                          # the later file wins, like dict `|`
    for name, content in (("a.json", half), ("b.json", graphs), ("c.json", override)):
        (tmp_path / name).write_text(json.dumps(content, indent=1))
    sg = GQASceneGraphs(token_lists=g8["token_lists"], scene_graph_files=[str(tmp_path / n) for n in ("a.json", "b.json", "c.json")])
    d = sg.query_and_translate("img0")
    same_graph(dict(x=d.x, edge_index=d.edge_index, edge_attr=d.edge_attr, x_bbox=d.x_bbox, added_sym_edge=d.added_sym_edge),
               g8["per_image"]["img5"], graphs["img5"], g8["stoi"])
    assert d.edge_attr.dim() == 2
    data = [(f"q{i}", k, f"what is {k}?", {"structural": "query"}, i, k) for i, k in enumerate(["img1", "img2", "nope"])]
    qid, batch, qs, qmask, labels, img, types = gqa_collate(data, sg, tokenizer=None, pin_memory=False)
    assert qid == ("q0", "q1", "q2") and labels.tolist() == [0, 1, 2] and batch.num_graphs == 3
    assert batch.x.size(0) == sum(g8["per_image"][k]["x"].size(0) for k in ("img1", "img2", "not-in-the-file"))


def _random_graphs(rng, n_graphs, names, attrs, rels):
    graphs = {}
    for g in range(n_graphs):
        n = rng.randint(0, 12)
        ids = rng.sample(range(1, 5000), n)
        objs = {}
        for oid in ids:
            o = {"name": rng.choice(names), "attributes": [rng.choice(attrs) for _ in range(rng.randint(0, 5))],
                 # (a 1-node graph with a self relation breaks the reference itself: x.squeeze() makes it 1-D, gqa.py:172)
                 "relations": [{"object": str(rng.choice(ids)), "name": rng.choice(rels)}
                               for _ in range(rng.randint(0, 4) if n > 1 else 0)],
                 "w": rng.random(), "nested": {"a": [1, 2, {"b": None}], "t": True, "f": False}}
            if rng.random() < 0.3:
                o.update(x1=rng.randint(-5, 600), y2=rng.randint(0, 600))
            objs[str(oid)] = o
        graphs[f"g{g}"] = {"objects": objs, "location": "indoors \"quoted\" \\ back\nslash", "width": 640}
    return graphs


def test_fuzz_against_oracle(L, g8):
    rng = random.Random(7)
    names = g8["token_lists"][0] + ["oov-name", "café", "emoji \U0001F600"]
    attrs = g8["token_lists"][1] + ["oov-attr"]
    rels = g8["token_lists"][2] + ["oov rel"]
    graphs = _random_graphs(rng, 300, names, attrs, rels)
    st = L.SceneGraphStore(L.SceneGraphVocab(g8["token_lists"]))
    st.add_json(json.dumps(graphs))                              # ensure_ascii: \\u escapes and surrogate pairs
    st2 = L.SceneGraphStore(L.SceneGraphVocab(g8["token_lists"]))
    st2.add_json(json.dumps(graphs, ensure_ascii=False, indent=2).encode("utf-8"))   # raw UTF-8, whitespace
    keys = list(graphs) + ["missing"]
    rng.shuffle(keys)
    ref = OL.collate([OL.dataset_item(OL.query_and_translate(graphs, k, g8["stoi"])) for k in keys])
    for s in (st, st2):
        b = s.collate(keys, pin_memory=False)
        for name in ("x", "edge_index", "edge_attr", "x_bbox", "added_sym_edge", "batch", "ptr"):
            assert torch.equal(getattr(b, name), ref[name]), name


def test_slots_threads_and_reused_buffers_give_the_same_batch(L, g8):
    rng = random.Random(11)
    graphs = _random_graphs(rng, 700, g8["token_lists"][0], g8["token_lists"][1], g8["token_lists"][2])
    st = L.SceneGraphStore(L.SceneGraphVocab(g8["token_lists"])).add_json(json.dumps(graphs))
    keys = [rng.choice(list(graphs) + ["nope"]) for _ in range(1500)]
    slots = st.slots(keys)
    assert int((slots < 0).sum()) == keys.count("nope")
    ref = st.collate(keys, pin_memory=False, threads=1)
    bufs = L.BatchBuffers(pin_memory=False)
    for threads in (1, 3, 8):
        st.collate(slots[:100], out=bufs, threads=threads)        # a smaller batch first: the buffers are re-used
        b = st.collate(slots, out=bufs, threads=threads)
        for name in ("x", "edge_index", "edge_attr", "x_bbox", "added_sym_edge", "batch", "ptr"):
            assert torch.equal(getattr(b, name), getattr(ref, name)), (name, threads)
        assert (b.max_nodes, b.max_edges) == (ref.max_nodes, ref.max_edges)


@pytest.mark.parametrize("text,fragment", [
    ('{"a": {"objects": {"1": {"name": "x", "attributes": [], "relations": [{"object": "2", "name": "on"}]}}}}', "unknown object"),
    ('{"a": {"objects": {"1": {"name": "x", "attributes": []}}}}', "lacks"),
    ('{"a": {"width": 3}}', "no 'objects'"),
    ('{"a": {"objects": {}}', "expected"),
    ('[1, 2]', "expected '{'"),
    ('{"a": {"objects": {}}} trailing', "trailing"),
    ('{"a": {"junk": ' + "[" * 200 + "]" * 200 + ', "objects": {}}}', "nesting too deep"),
])
def test_malformed_input_fails_loudly(L, g8, text, fragment):
    st = L.SceneGraphStore(L.SceneGraphVocab(g8["token_lists"]))
    with pytest.raises(L.LoaderError) as e:
        st.add_json(text)
    assert fragment in str(e.value)
    with pytest.raises(L.LoaderError):
        st.add_json_file("/nonexistent/file.json")


def test_number_at_the_very_end_of_an_unterminated_buffer(L, g8):
    """The parser gets (pointer, length): a number that ends the buffer must be read from inside it (strtod on
    the raw pointer).  The text is passed as a slice of a larger bytes object whose next bytes are digits."""
    st = L.SceneGraphStore(L.SceneGraphVocab(g8["token_lists"]))
    good = '{"a": {"width": 12, "objects": {}}}'
    st.add_json(good)
    with pytest.raises(L.LoaderError):           # truncated right after the number: must fail inside the view, not read on
        st.add_json('{"a": {"width": 12')


def test_collate_hands_over_per_graph_sizes_when_a_graph_lies_beyond_a_tile(L, g8):
    """The reference caps nothing (datasets/scene_graph.py:199-389): a scene graph with 70 objects is beyond the 64-node tile of the
    GPU kernels.  collate then carries graph_sizes = (nodes, in-edges) per graph on the HOST (ops.GraphPlan.build's hint: the list of
    such graphs is made without a device-to-host sync); a batch of small graphs carries none."""
    rng = random.Random(11)
    names, attrs, rels = g8["token_lists"][0], g8["token_lists"][1], g8["token_lists"][2]
    graphs = _random_graphs(rng, 20, names, attrs, rels)
    ids = list(range(1, 71))
    graphs["big"] = {"objects": {str(i): {"name": rng.choice(names), "attributes": [],
                                          "relations": [{"object": str(rng.choice(ids)), "name": rng.choice(rels)} for _ in range(2)]}
                                 for i in ids}, "width": 640}
    st = L.SceneGraphStore(L.SceneGraphVocab(g8["token_lists"]))
    st.add_json(json.dumps(graphs))
    small = st.collate([k for k in graphs if k != "big"], pin_memory=False)
    assert small.graph_sizes is None and small.max_nodes <= 64
    keys = list(graphs)
    b = st.collate(keys, pin_memory=False)
    assert b.max_nodes == 70 and b.graph_sizes is not None and b.graph_sizes.device.type == "cpu"
    assert tuple(b.graph_sizes.shape) == (2, len(keys))
    assert torch.equal(b.graph_sizes[0], b.ptr[1:] - b.ptr[:-1])
    assert torch.equal(b.graph_sizes[1], torch.bincount(b.batch[b.edge_index[1]], minlength=len(keys)))
    assert int(b.graph_sizes[0].max()) == b.max_nodes and int(b.graph_sizes[1].max()) == b.max_edges
    moved = b.to("cpu")
    assert moved.graph_sizes is b.graph_sizes
