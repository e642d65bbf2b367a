#!/usr/bin/env python3
"""Build libisg_hip.so from the WORKING TREE's sources with extra compiler flags into tools/_build/, for same-process A/Bs with
tools/ab_libs.py.   python3 tools/build_variant.py <name> [flags for every file ...] [file.hip:flag ...]
e.g.  python3 tools/build_variant.py noslp isg_layer_tile.hip:-fno-slp-vectorize   ->  tools/_build/libisg_noslp.so"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIP_FLAGS, EXTRA_FLAGS

name = sys.argv[1]
every = [a for a in sys.argv[2:] if a.startswith("-")]
per = {}
for a in sys.argv[2:]:
    if not a.startswith("-"):
        f, flag = a.split(":", 1)
        per.setdefault(f, []).append(flag)
csrc = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
out = os.path.join(ROOT, "tools", "_build")
os.makedirs(out, exist_ok=True)
srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
with tempfile.TemporaryDirectory(prefix="isg_var_") as tmp:
    def cc(f):
        o = os.path.join(tmp, f[:-4] + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *HIP_FLAGS, *EXTRA_FLAGS.get(f, []), *every, *per.get(f, []), "-c", f, "-o", o], cwd=csrc)
        return o

    with ThreadPoolExecutor(max_workers=7) as pool:
        objs = list(pool.map(cc, srcs))
    lib = os.path.join(out, f"libisg_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", lib])
    print(lib)
# a variant library is measured against the shipped one: it has to obey the same rule about packed fp32 operands
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "tools", "scan_pk_cross.py"), lib]))
