#!/usr/bin/env python3
"""Kernel statistics (calls, total / average / min / max duration, share) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes DIR/NAME_results.db on ROCm 7.2), as CSV on stdout.
  python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--skip N]   (--skip: drop each kernel's first N calls)"""
import re
import sqlite3
import sys

path = sys.argv[1]
skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
db = sqlite3.connect(path)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
by = {}
for name, s, e in rows:
    by.setdefault(name, []).append((e - s) / 1e3)
tot_all = sum(sum(v[skip:]) for v in by.values())
print("Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage")
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1][skip:])):
    v = v[skip:]
    if not v:
        continue
    short = re.sub(r"\s+", " ", name)
    print(f"\"{short}\",{len(v)},{sum(v):.1f},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f},{100 * sum(v) / tot_all:.2f}")
