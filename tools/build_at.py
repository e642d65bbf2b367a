#!/usr/bin/env python3
"""Build libisg_hip.so from the sources of another commit (or of the working tree with one file swapped) into tools/_build/, for
same-process A/Bs with tools/ab_libs.py.   python3 tools/build_at.py <commit> <name>     -> tools/_build/libisg_<name>.so"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIP_FLAGS, EXTRA_FLAGS

commit, name = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "tools", "_build")
os.makedirs(out, exist_ok=True)
tmp = tempfile.mkdtemp(prefix="isg_at_")
try:
    subprocess.check_call(f"git -C {ROOT} archive {commit} intrinsic-subgraph-generation-for-vqa_amd/csrc include | tar -x -C {tmp}", shell=True)
    csrc = os.path.join(tmp, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
    srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))

    def cc(f):
        subprocess.check_call(["/opt/rocm/bin/hipcc", *HIP_FLAGS, *EXTRA_FLAGS.get(f, []), "-c", f, "-o", f[:-4] + ".o"], cwd=csrc)
        return os.path.join(csrc, f[:-4] + ".o")

    with ThreadPoolExecutor(max_workers=7) as pool:
        objs = list(pool.map(cc, srcs))
    lib = os.path.join(out, f"libisg_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", lib])
    print(lib)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
