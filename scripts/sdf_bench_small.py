"""Latency of the 16-point-tile fused SDF kernel at small batches (profiling helper)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch, bench
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
torch.manual_seed(0)
net = IDRNetwork(bench.idr_conf("C2")).cuda().implicit_network
for n, tile in ((16, 4), (480, 4), (1024, 4), (480, 8), (1024, 8), (2048, 8), (2048, 16), (4096, 16)):
    net.sdf_tile_points = tile
    x = torch.rand(n, 3, device="cuda") * 2 - 1
    for _ in range(3):
        net.sdf(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 50
    s.record()
    for _ in range(it):
        net.sdf(x)
    e.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('HM_LIB_PATH','default')} n={n:6d} tile={tile:2d} {s.elapsed_time(e) / it * 1e3:9.1f} us")
