"""Throughput of the C++ scene-graph loader vs the Python restatement of the reference's conversion + collate
(oracle/loader.py, timed as the CPU baseline) on synthetic GQA-shaped scene graphs.
Usage: python tools/bench_loader.py [--images 20000] [--batch 4096]"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isubgvqa_amd  # noqa: E402,F401
from isubgvqa_amd import loader  # noqa: E402


def synth(n_images, rng):
    names = [f"name{i}" for i in range(1700)]
    attrs = [f"attr{i}" for i in range(600)]
    rels = [f"rel {i}" for i in range(300)]
    graphs = {}
    for g in range(n_images):
        n = max(2, int(rng.gauss(17, 6)))
        ids = [str(rng.randint(100000, 4999999)) for _ in range(n)]
        objs = {}
        for oid in ids:
            objs[oid] = {"name": rng.choice(names), "h": rng.randint(5, 300), "w": rng.randint(5, 300),
                         "x": rng.randint(0, 500), "y": rng.randint(0, 400),
                         "attributes": [rng.choice(attrs) for _ in range(rng.randint(0, 3))],
                         "relations": [{"object": rng.choice(ids), "name": rng.choice(rels)} for _ in range(rng.randint(0, 5))]}
        graphs[str(2300000 + g)] = {"width": 500, "height": 375, "location": "outdoors", "objects": objs}
    return graphs, [names, attrs, rels, [], [], []]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--oracle-images", type=int, default=4096)
    a = ap.parse_args()
    rng = random.Random(0)
    graphs, token_lists = synth(a.images, rng)
    text = json.dumps(graphs).encode()
    print(f"{a.images} images, {len(text) / 1e6:.1f} MB of JSON")

    t0 = time.perf_counter()
    py = json.loads(text)
    t_json = time.perf_counter() - t0

    vocab = loader.SceneGraphVocab(token_lists)
    t0 = time.perf_counter()
    store = loader.SceneGraphStore(vocab).add_json(text)
    t_load = time.perf_counter() - t0
    print(f"C++ parse + convert: {t_load:.3f} s = {len(text) / 1e6 / t_load:.0f} MB/s = {a.images / t_load:,.0f} images/s "
          f"(python json.loads alone: {t_json:.3f} s)")

    keys = list(graphs)
    batches = [[rng.choice(keys) for _ in range(a.batch)] for _ in range(8)]
    t0 = time.perf_counter()
    slot_batches = [store.slots(b) for b in batches]          # a dataset does this once, at construction
    t_slots = (time.perf_counter() - t0) / len(batches)
    for threads in (1, 2, 4, 8):
        bufs = loader.BatchBuffers(pin_memory=torch.cuda.is_available())
        store.collate(slot_batches[0], out=bufs, threads=threads)
        store.collate(slot_batches[1], out=bufs, threads=threads)
        t0 = time.perf_counter()
        for _ in range(5):
            for b in slot_batches:
                out = store.collate(b, out=bufs, threads=threads)
        t_col = (time.perf_counter() - t0) / (5 * len(batches))
        nbytes = 8 * (out.x.numel() * 2 + out.edge_index.numel() + out.edge_attr.numel() + out.batch.numel())
        print(f"C++ collate of {a.batch} graphs, {threads} thread(s), reused buffers: {t_col * 1e3:.2f} ms = "
              f"{a.batch / t_col:,.0f} graphs/s = {nbytes / t_col / 1e9:.1f} GB/s written "
              f"(N={out.x.size(0)}, E={out.edge_index.size(1)}; id->slot lookup {t_slots * 1e3:.2f} ms once per dataset)")

    from oracle import loader as OL      # CPU baseline: the reference's per-image conversion + Batch.from_data_list
    stoi = vocab.get_stoi()
    sample = batches[0][:a.oracle_images]
    t0 = time.perf_counter()
    items = [OL.dataset_item(OL.query_and_translate(py, k, stoi)) for k in sample]
    t_conv = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = OL.collate(items)
    t_pc = time.perf_counter() - t0
    got = store.collate(sample, pin_memory=False)
    assert torch.equal(got.edge_index, ref["edge_index"]) and torch.equal(got.edge_attr, ref["edge_attr"])
    print(f"python port: convert {len(sample) / t_conv:,.0f} images/s (first access), collate {len(sample) / t_pc:,.0f} graphs/s "
          f"-> C++ collate is {t_pc / len(sample) * a.batch / t_col:.0f}x the cached-python collate, "
          f"{(t_conv + t_pc) / len(sample) * a.batch / t_col:.0f}x a cold batch")


if __name__ == "__main__":
    main()
