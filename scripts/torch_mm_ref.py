"""Reference point only (NOT used by the product): how fast is the vendor library on the training GEMM shapes?"""
import torch
for (M, N, K) in [(1750, 512, 512), (3072, 512, 512), (4822, 512, 512), (65536, 512, 512)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    for _ in range(5): c = a @ b.t()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): c = a @ b.t()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"torch.mm M={M} N={N} K={K}: {us:.1f} us  {2*M*N*K/us/1e6:.1f} TF")
