import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isubgvqa_amd
from isubgvqa_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
libs = {k: ctypes.CDLL(os.path.join(here, f"_ab_{k}_gemm.so")) for k in ("old", "new")}
for l in libs.values():
    l.isg_linear_bf16x6.restype = ctypes.c_int
    l.isg_linear_bf16x6.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int32] * 5 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
shapes = [(82286, 128, 1024, 0), (205024, 128, 512, 0), (82286, 512, 256, 1), (82286, 256, 128, 1), (82286, 128, 128, 1), (4096, 512, 1842, 0)]
st = torch.cuda.current_stream().cuda_stream
for M, K, N, act in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    planes = ops._weight_planes(w, False)
    outs = {k: torch.empty(M, N, device=dev) for k in libs}
    res = {k: [] for k in libs}
    for r in range(14):
        for k in libs:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = libs[k].isg_linear_bf16x6(x.data_ptr(), planes.data_ptr(), b.data_ptr(), outs[k].data_ptr(), M, N, K, K, N, act, st)
            e.record(); torch.cuda.synchronize()
            assert rc == 0
            if r >= 2: res[k].append(s.elapsed_time(e) * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    same = torch.equal(outs["old"], outs["new"])
    print(f"M={M} K={K} N={N} act={act}: old {med['old']:.1f} us  new {med['new']:.1f} us  ({med['old']/med['new']:.3f}x) identical={same}")
