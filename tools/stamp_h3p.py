#!/usr/bin/env python3
"""Where does a wave of the persistent planes32 GEMM (isg_linear_h3p, linear_h3q_kernel) spend its cycles?  `--build` (in the
build container) makes tools/_build/libisg_p3_stamp.so with -DISG_DIAG; the run launches it on a few shapes and prints the
mean core-clock cycles per segment and k-tile over all waves (s_memtime; a stamp waits for the LDS reads before it)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "libisg_p3_stamp.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DISG_DIAG",
                           *[a for a in sys.argv[1:] if a.startswith("-D")],
                           os.path.join(CSRC, "isg_gemm_h3p.hip"), os.path.join(CSRC, "isg_graph.hip"), "-o", OUT])
    print("built", OUT)
    sys.exit(0)

import torch

from isubgvqa_amd import _lib, ops

stamp = ctypes.CDLL(OUT)
stamp.isg_linear_h3p.restype, stamp.isg_linear_h3p.argtypes = _lib.SIGNATURES["isg_linear_h3p"]
stamp.isg_p3_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
NAMES = ["L0: 12 reads + 2 requests (issue + LDS latency)", "L0: vmcnt wait", "L0: barrier", "M0: lgkmcnt wait", "M0: 24 MFMAs (+ piece) issue",
         "M0: barrier", "L1: (piece, when it sat here)", "L1: 4 reads + 4 requests + stager", "L1: vmcnt wait", "L1: barrier", "M1: lgkmcnt wait",
         "M1: 24 MFMAs (+ piece) issue", "M1: barrier", "L1: piece: register select (switch)", "L1: piece: parameter reads from LDS", "(total)"]
SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a and not a.startswith("-")] or [(49152, 1536, 512), (65536, 1024, 2048)]
if "--lib" in sys.argv:           # another stamped build (e.g. of an older source), for a same-box comparison
    OUT = sys.argv[sys.argv.index("--lib") + 1]
    stamp = ctypes.CDLL(OUT)
    stamp.isg_linear_h3p.restype, stamp.isg_linear_h3p.argtypes = _lib.SIGNATURES["isg_linear_h3p"]
    stamp.isg_p3_set_stamp_buffer.argtypes = [ctypes.c_void_p]
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    xp = ops.split_planes32(x)
    wp, winv, bound = ops._h3p_weight(w, b, False)
    out = torch.empty(M, N, device=dev)
    buf = torch.zeros(256 * 8, 16, dtype=torch.int64, device=dev)
    assert stamp.isg_p3_set_stamp_buffer(buf.data_ptr()) == 0
    for rep in range(2):
        buf.zero_()
        rc = stamp.isg_linear_h3p(xp.planes.data_ptr(), xp.inv.data_ptr(), wp.data_ptr(), winv.data_ptr(), b.data_ptr(), out.data_ptr(),
                                  0, 0, 0, M, N, K, N, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
    s = buf.double().cpu()
    s = s[s[:, 15] > 0]
    tiles = ((M + 255) // 256) * ((N + 127) // 128)
    kts = tiles * ((K + 31) // 32) / 256.0
    tot = s[:, 15].mean().item()
    print(f"{M} x {N} x {K}: {tiles} tiles, {kts:.0f} k-tiles per workgroup; a wave lives {tot:.0f} cycles = {tot / kts:.0f} per k-tile "
          f"(MFMA issue alone: 768)")
    for grp, sel in (("group 0 (waves 0-3)", [w for w in range(s.size(0)) if (w % 8) < 4]), ("group 1 (waves 4-7)", [w for w in range(s.size(0)) if (w % 8) >= 4])):
        print(" ", grp)
        for i, n in enumerate(NAMES):
            v = s[sel, i].mean().item()
            print(f"    {n:55s} {v / kts:8.1f} per k-tile  ({100 * v / tot:5.1f} %)")
