"""hm_gemm_f32 on the chain shapes (N = K = 512) as a function of the row count M, 20 launches replayed from a HIP
graph: how much would a grad-path evaluation on fewer rows (surface rays only) save?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
import torch
from hashmodnffbanks_idr_amd import ops


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M in (512, 896, 1024, 1536, 2048, 3072, 4096):
    a = torch.randn(M, 512, device="cuda"); w = torch.randn(512, 512, device="cuda"); bias = torch.zeros(512, device="cuda")
    out = torch.empty(M, 512, device="cuda")
    nt = timed(lambda: ops.gemm_ep(a, w, bias, False, True, ops.EPI_SOFTPLUS, 100.0, 20.0))
    nn = timed(lambda: ops.gemm(a, w, None, False, False, out=out))
    u = torch.randn(2 * M, 512, device="cuda"); v = torch.randn(2 * M, 512, device="cuda"); dW = torch.zeros(512, 512, device="cuda")
    tn = timed(lambda: ops.gemm(u, v, None, True, False, out=dW, accumulate=True))
    print(f"M={M:5d}: X W^T+softplus {nt:6.1f} us | dY W {nn:6.1f} us | dW (K=2M) {tn:6.1f} us")
