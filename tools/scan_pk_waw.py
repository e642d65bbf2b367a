#!/usr/bin/env python3
"""Scan the compiled kernels for the instruction pattern behind isg_gatv2_tile_conv's intermittent wrong sums:
two packed fp32 operations (v_pk_fma / v_pk_mul / v_pk_add _f32) writing the SAME register pair within a few issue slots of each
other with DIFFERENT operand selection, one of them taking a low-half operand from a high dword (op_sel:[..1..]); mixing the
op_sel_hi forms alone is everywhere (7700 pairs) in kernels that never differed.   python3 tools/scan_pk_waw.py [window]"""
import collections
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
window = int(sys.argv[1]) if len(sys.argv) > 1 else 6
out = "/tmp/isg_scan"
os.makedirs(out, exist_ok=True)
hits = 0
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    asm = os.path.join(out, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", src, "-o", asm],
                   check=True, stderr=subprocess.DEVNULL)
    kern, recent = None, collections.deque(maxlen=window)
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, recent = m.group(1), collections.deque(maxlen=window)
            continue
        t = line.strip()
        if not line.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        if op.startswith("s_waitcnt") or op == "s_nop":
            continue
        pk = re.match(r"(v_pk_(?:fma|mul|add)_f32) (v\[\d+:\d+\]),", t)
        if pk:
            sel = " ".join(re.findall(r"op_sel(?:_hi)?:\[[\d,]+\]", t))
            for prev_dst, prev_sel, prev_t in recent:
                cross = lambda z: bool(re.search(r"op_sel:\[[\d,]*1", z))      # a LOW-half operand taken from a high dword
                if prev_dst == pk.group(2) and prev_sel != sel and (cross(sel) or cross(prev_sel)):
                    hits += 1
                    print(f"{os.path.basename(src)}: {kern[:70]}\n    {prev_t}\n    {t}")
            recent.append((pk.group(2), sel, t))
        else:
            recent.append((None, None, t))
print(f"{hits} pair(s) of packed fp32 operations on one register pair with different operand selection within {window} instructions")
