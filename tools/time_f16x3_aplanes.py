#!/usr/bin/env python3
"""Upper bound of handing activations to the fp16 three-product tile kernel PRE-SPLIT: the shipped library converts fp32 A rows
to (hi, mid) planes inside every tile; tools/_build/libisg_hip_aplanes.so (the library built with -DISG_F16X3_APLANES) reads
`a` as planes [2][M][K] and stages them with 16-byte LDS stores, no conversion.  The planes are REAL ones (isg_split_f16x2_rows
of the same A: operand data decides the clock the chip holds under MFMA load, so random bits would not do); only the time
counts, the diagnostic's row scales are not those of the planes.   python3 tools/time_f16x3_aplanes.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from isubgvqa_amd import _lib

c = ctypes
SIG = [c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64,
       c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_void_p]
libs = {"shipped (fp32 A, split in the tile loop)": ctypes.CDLL(_lib.LIB_PATH),
        "diagnostic (A as two separate planes)": ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libisg_hip_aplanes.so")),
        "diagnostic (A planes interleaved per k-tile)": ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libisg_hip_aplanes2.so")),
        "diagnostic (interleaved, shipped thread map)": ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libisg_hip_aplanes3.so"))}
for lib in libs.values():
    lib.isg_linear_f16x3_tile.argtypes = SIG
    lib.isg_split_f16x2_rows.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(1 << 27, device=dev)
shapes = [("text in_proj", 49152, 1536, 512), ("text out_proj / q", 49152, 512, 512), ("x_proj.0 cfg2", 82286, 256, 512),
          ("lin_l|r C=300 (K padded to 320)", 82189, 2400, 320)]
ref = next(iter(libs.values()))
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    wp, winv = torch.empty(2, N, K, dtype=torch.int16, device=dev), torch.empty(N, device=dev)
    assert ref.isg_split_f16x2_rows(w.data_ptr(), N, K, wp.data_ptr(), winv.data_ptr(), st) == 0
    xp, xinv = torch.empty(2, M, K, dtype=torch.int16, device=dev), torch.empty(M, device=dev)
    assert ref.isg_split_f16x2_rows(x.data_ptr(), M, K, xp.data_ptr(), xinv.data_ptr(), st) == 0        # real planes of A
    xi = xp.view(2, M, K // 32, 32).permute(1, 2, 0, 3).contiguous()      # [M][K/32][hi 32 | mid 32]: one line per row and k-tile
    rm = x.abs().amax(1, keepdim=True).contiguous()
    d = torch.empty(M, N, device=dev)
    res = {k: [] for k in libs}
    for r in range(12):
        for label, lib in libs.items():
            a_ptr = xi.data_ptr() if "interleaved" in label else (xp.data_ptr() if label.startswith("diag") else x.data_ptr())
            flush.fill_(float(r))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = lib.isg_linear_f16x3_tile(a_ptr, rm.data_ptr(), 1, 1, wp.data_ptr(), winv.data_ptr(), b.data_ptr(), d.data_ptr(),
                                           None, M, N, K, K, N, 0, K, 0, 0, st)
            e.record()
            torch.cuda.synchronize()
            assert rc == 0, rc
            if r >= 2:
                res[label].append(s.elapsed_time(e) * 1e3)
    print(f"{name} [{M} x {N} x {K}]")
    for label, v in res.items():
        v = sorted(v)
        print(f"    {label:44s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us")
