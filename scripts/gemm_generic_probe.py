"""Which grad-path GEMMs miss the pipelined kernel, and what do they cost?  (HM_GEMM_LOG=1 lists them on stderr.)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch
from hashmodnffbanks_idr_amd import ops

def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3

for rows in (3072, 2048):
    x67 = torch.randn(rows, 67, device="cuda"); w0 = torch.randn(512, 67, device="cuda"); b = torch.zeros(512, device="cuda")
    u = torch.randn(rows, 512, device="cuda"); u445 = torch.randn(rows, 445, device="cuda"); w3 = torch.randn(445, 512, device="cuda")
    z = torch.randn(rows, 512, device="cuda")
    print(rows, "fwd l0  X[.,67] W0^T + softplus", round(t(lambda: ops.gemm_ep(x67, w0, b, False, True, ops.EPI_SOFTPLUS, 100.0, 20.0)), 1), "us")
    print(rows, "rev l0  u[.,512] W0 -> [.,67]   ", round(t(lambda: ops.gemm(u, w0, None, False, False)), 1), "us")
    print(rows, "rev l3  u[.,445] W3 (S1MUL)     ", round(t(lambda: ops.gemm_ep(u445, w3, None, False, False, ops.EPI_S1MUL, 100.0, 20.0, z=z, nz=512)), 1), "us")
    print(rows, "fwd l3  X[.,512] W3^T + softplus", round(t(lambda: ops.gemm_ep(u, w3, torch.zeros(445, device='cuda'), False, True, ops.EPI_SOFTPLUS, 100.0, 20.0)), 1), "us")
    print(rows, "ref     X[.,512] W^T + softplus ", round(t(lambda: ops.gemm_ep(u, torch.randn(512, 512, device='cuda'), b, False, True, ops.EPI_SOFTPLUS, 100.0, 20.0)), 1), "us")
