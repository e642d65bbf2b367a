#!/usr/bin/env python3
"""Where does isg_linear_panel spend its time?  Builds nothing: expects tools/_build/libisg_panel_abl.so (made by
`python tools/ablate_panel.py --build` in the build container: isg_gemm_panel.hip with -DISG_PANEL_ABLATION) and times
compile-time ablated variants of the kernel, selected by ISG_PANEL_DBG, interleaved in one process with HIP events.
  DBG bits: 1 no B loads in the k loop, 2 no epilogue stores, 4 no MFMA, 8 no A staging, 16 no A-fragment LDS reads."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "intrinsic-subgraph-generation-for-vqa_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build", "libisg_panel_abl.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                           "-DISG_PANEL_ABLATION", os.path.join(CSRC, "isg_gemm_panel.hip"),
                           os.path.join(CSRC, "isg_graph.hip"), "-o", OUT])
    print("built", OUT)
    sys.exit(0)

import torch

lib = ctypes.CDLL(OUT)
c = ctypes
lib.isg_split_bf16x3_frag_elems.restype = c.c_int64
lib.isg_split_bf16x3_frag_elems.argtypes = [c.c_int64, c.c_int32]
lib.isg_split_bf16x3_frag.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p]
lib.isg_linear_panel.argtypes = [c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int64, c.c_int32,
                                 c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_void_p]
dev = torch.device("cuda:0")
shapes = [("lin_edge", 205024, 128, 512), ("lin_l|r", 82286, 128, 1024), ("x_proj.0", 82286, 512, 256)]
variants = [(0, "full"), (2, "no stores"), (1, "no B loads"), (16, "no A frag reads"), (8, "no A staging"), (4, "no MFMA"),
            (6, "no MFMA, no stores"), (3, "no B loads, no stores"), (27, "MFMA only (no loads/stores)"),
            (31, "nothing (launch + barriers)")]
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    planes = torch.empty(int(lib.isg_split_bf16x3_frag_elems(N, K)), dtype=torch.int16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.isg_split_bf16x3_frag(x.new_tensor(w).data_ptr(), N, K, planes.data_ptr(), st) == 0
    out = torch.empty(M, N, device=dev)
    res = {v: [] for v, _ in variants}
    for r in range(8):
        for v, _ in variants:
            os.environ["ISG_PANEL_DBG"] = str(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = lib.isg_linear_panel(x.data_ptr(), 0, planes.data_ptr(), None, out.data_ptr(), 0, M, N, K, K, N, 0, st)
            e.record()
            torch.cuda.synchronize()
            assert rc == 0
            if r >= 2:
                res[v].append(s.elapsed_time(e) * 1e3)
    print(f"{name} [{M}x{K}]x[{N}x{K}]")
    for v, label in variants:
        t = sorted(res[v])[len(res[v]) // 2]
        print(f"   DBG={v:2d} {label:32s} {t:8.1f} us", flush=True)
