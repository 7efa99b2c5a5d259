"""fixed cost of one hm_gemm_f32 launch: K sweep at M=2048, N=512 under graph replay (profiling helper)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
import torch
from hashmodnffbanks_idr_amd import ops
sys.path.insert(0, R + "/scripts")
from gemm_graph_bench import timed  # noqa: E402  (prints its own table first)
print("---- K sweep")
for K in (32, 64, 128, 256, 512, 1024):
    a = torch.randn(2048, K, device="cuda"); b = torch.randn(512, K, device="cuda"); out = torch.empty(2048, 512, device="cuda")
    t = timed(lambda: ops.gemm(a, b, None, False, True, out=out))
    v = timed(lambda: torch.mm(a, b.t(), out=out))
    print(f"K={K:5d}: hm {t:6.1f} us | vendor {v:6.1f} us")
x = torch.empty(2048 * 512, device="cuda")
print("fill kernel (1M floats):", round(timed(lambda: x.fill_(1.0)), 2), "us")
