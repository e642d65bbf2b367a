#!/usr/bin/env python3
"""Average per launch of every counter in rocprofv3 --pmc output directories, for kernels whose name contains a substring.
    python3 tools/pmc_sum.py <kernel-substring> <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

kern = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:32s} {sum(v) / len(v) / 1e6:12.3f} M   ({len(v)} launches)")
