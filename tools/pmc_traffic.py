#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/profile_mp.py into the per-launch HBM traffic
of the message-passing kernel, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950:
  * both counters are in KiB;
  * FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) streaming reads at 64 B -> multiply by 2; the factor is
    re-derived here from the known-size copy in the same run (1 GiB read, 1 GiB written);
  * WRITE_SIZE is exact for 16 B/lane stores (checked against the same copy).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [kernel-substring[+kernel-substring]]
A "+"-joined pair (gatv2_edge_logits+gatv2_mp_graph) sums the two kernels' per-launch averages: the edge-logits pair."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    out = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def calibration_copies(table, expected_kib):
    """Counter values of the 1 GiB calibration copies only.  Any other device-to-device copy of the run (a few-KiB
    copyBuffer dispatch showed up in round 2 and, averaged in, turned the scales into 4/3 of their value) is dropped:
    a calibration copy moves within a factor of 4 of the known size, everything else is reported as excluded."""
    vals = [v for k, vs in table.items() if "copyBuffer" in k for v in vs]
    keep = [v for v in vals if expected_kib / 4 <= v <= expected_kib * 4]
    if not keep:
        raise SystemExit(f"no calibration copy near {expected_kib} KiB among {vals}")
    return keep, [v for v in vals if v not in keep]


def main():
    fetch_dir, write_dir, out_path = sys.argv[1:4]
    kern = sys.argv[4] if len(sys.argv) > 4 else "gatv2_mp"
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    gib_kib = 1 << 20
    copy_f, drop_f = calibration_copies(fetch, gib_kib // 2)  # FETCH_SIZE counts the 1 GiB read as ~512 Ki KiB
    copy_w, drop_w = calibration_copies(write, gib_kib)
    f_scale = gib_kib / (sum(copy_f) / len(copy_f))          # expected 2.0 on gfx950
    w_scale = gib_kib / (sum(copy_w) / len(copy_w))          # expected 1.0
    # the guide PRESCRIBES x2 / x1; the copy is a cross-check of that, and bench.py replays this file into
    # roofline.traffic, so a calibration that disagrees stops here instead of being written out
    if abs(f_scale - 2.0) > 0.1 or abs(w_scale - 1.0) > 0.05:
        raise SystemExit(f"calibration off the guide's corrections: fetch x{f_scale:.4f} (2.0), write x{w_scale:.4f} "
                         f"(1.0); copies used {copy_f} / {copy_w}")
    if "+" in kern:
        parts = kern.split("+")
        per = []
        for tab in (fetch, write):
            cols = [[v for k, vs in tab.items() if p_ in k for v in vs] for p_ in parts]
            n = min(len(c) for c in cols)
            per.append([sum(c[i] for c in cols) for i in range(n)])      # launch i of every part, in launch order
        kf, kw = per
        name = " + ".join(next(k.split("(")[0] for k in fetch if p_ in k) for p_ in parts) + "("
    else:
        kf = [v for k, vs in fetch.items() if kern in k for v in vs]
        kw = [v for k, vs in write.items() if kern in k for v in vs]
        name = [k for k in fetch if kern in k][0]
    half = len(kf) // 2                                      # profile_mp.py: first half unmasked, second half masked
    # the log of THIS fetch pass (rocprofv3 ... -d <dir> ... > <dir>.log); never a sibling's: a neighbour's log once labelled
    # the edge-logits pair's traffic as the un-fused kernel's
    import os
    log_path = fetch_dir.rstrip("/") + ".log"
    log = open(log_path).read() if os.path.exists(log_path) else ""
    m = re.search(r"N=(\d+) E=(\d+) H=(\d+) C=(\d+) bytes_unmasked=(\d+) bytes_masked=(\d+)", log)
    own = re.search(r"pair_own_bytes_unmasked=(\d+)", log)
    res = {
        "kernel": name.split("(")[0],
        "fetch_kib_raw": sum(kf[:half]) / half, "write_kib_raw": sum(kw[:half]) / half,
        "fetch_scale_from_copy": round(f_scale, 4), "write_scale_from_copy": round(w_scale, 4),
        "calibration_copies_used": len(copy_f), "other_copies_excluded_kib": {"fetch": drop_f, "write": drop_w},
        "hbm_bytes_per_launch": int((sum(kf[:half]) / half * f_scale + sum(kw[:half]) / half * w_scale) * 1024),
        "hbm_bytes_per_launch_masked": int((sum(kf[half:]) / (len(kf) - half) * f_scale +
                                            sum(kw[half:]) / (len(kw) - half) * w_scale) * 1024),
    }
    if m:
        res.update(N=int(m.group(1)), E=int(m.group(2)), H=int(m.group(3)), C=int(m.group(4)),
                   algorithmic_bytes=int(m.group(5)), algorithmic_bytes_masked=int(m.group(6)))
        res["traffic_over_algorithmic"] = round(res["hbm_bytes_per_launch"] / res["algorithmic_bytes"], 3)
    # what the summary describes is decided by the kernels that were summed, not by what a log happens to say
    res["kind"] = "logits_pair" if "+" in kern else ("layer_conv" if "layer_conv" in name else "tile_conv" if "tile_conv" in name else ("graph" if "graph" in name else "chunk"))
    if "+" in kern and not own:
        raise SystemExit(f"{log_path}: no pair_own_bytes_unmasked line -- not a `profile_mp.py ... logits` run")
    if own:      # the pair never touches e_proj: its own minimum is far below the un-fused bytes_mp
        res["pair_own_algorithmic_bytes"] = int(own.group(1))
        res["traffic_over_own_algorithmic"] = round(res["hbm_bytes_per_launch"] / int(own.group(1)), 3)
    # the bytes of the kernel source(s) this summary was measured on: bench.py's load_traffic refuses a summary whose
    # kernel has been edited since
    import os as _os
    import sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    import bench as _bench
    res["source"] = list(_bench.TRAFFIC_SOURCES[res["kind"]])
    res["source_sha256"] = _bench.kernel_source_hash(res["kind"])
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
