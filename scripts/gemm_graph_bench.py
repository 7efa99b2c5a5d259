"""hm_gemm_f32 vs the vendor library on the training GEMM shapes, both replayed from a HIP graph (20 back-to-back
launches, so host launch cost is out of the picture).  Reference point only - the product uses hm_gemm_f32."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
import torch
from hashmodnffbanks_idr_amd import ops


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        for _ in range(20):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 * 1e3


for (M, N, K, ta, tb, name) in [(2048, 512, 512, False, True, "fwd X W^T"), (3072, 512, 512, False, True, "fwd X W^T"),
                                (2048, 512, 512, False, False, "dX = dY W"), (3072, 512, 512, False, False, "dX = dY W"),
                                (512, 512, 2048, True, False, "dW = dY^T X"), (512, 512, 3072, True, False, "dW = dY^T X")]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    ours = timed(lambda: ops.gemm(a, b, None, ta, tb, out=out))
    aa, bb = (a.t() if ta else a), (b.t() if tb else b)
    vend = timed(lambda: torch.mm(aa, bb, out=out))
    fl = 2 * M * N * K
    print(f"{name:12s} M={M:5d} N={N:4d} K={K:5d}: hm {ours:6.1f} us {fl/ours/1e6:6.1f} TF | vendor {vend:6.1f} us {fl/vend/1e6:6.1f} TF")
