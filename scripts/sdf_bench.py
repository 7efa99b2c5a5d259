"""Latency of the fused SDF kernel at the batch sizes the ray tracer issues (profiling helper)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch, bench
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
torch.manual_seed(0)
net = IDRNetwork(bench.idr_conf("C2")).cuda().implicit_network
for n in (16, 480, 2048, 4096, 8192, 16384, 48000, 110000, 262144):
    x = torch.rand(n, 3, device="cuda") * 2 - 1
    for tile in (16, 64):
        if tile == 16 and n > 20000:
            continue
        net.sdf_tile_points = tile
        for _ in range(3):
            net.sdf(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20
        s.record()
        for _ in range(it):
            net.sdf(x)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / it * 1e3
        print(f"n={n:7d} tile={tile:2d}  {us:9.1f} us  {n/us:8.2f} Mpts/s  {n*3.67e6/us/1e6:7.1f} TF")
