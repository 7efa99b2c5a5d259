"""Per-CU weight-stream rate of the 4-point SDF body against the size of the network: 512-wide layers = 7.9 MB of packed
weights (more than an XCD's 4 MiB L2: every round streams them from the Infinity Cache), 352-wide = 3.7 MB, 256-wide =
2.0 MB (L2 resident).  If the narrower networks stream faster per byte, the sparse rounds are paced by the L2 miss
latency, and filling the L2 ahead of the demand loads would pay."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import numpy as np, torch
from helpers import make_implicit
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand((256, 3), generator=g) * 2 - 1).cuda()
for width in (512, 384, 352, 256, 128):
    net = make_implicit("C2", (width,) * 8, 256, 3, 0.1, 0.05)
    net.eval()
    E = 67
    nbytes = 4 * (E * width + 6 * width * width + (width - E) * width - 0 + (width) * width + width)  # approx
    nbytes = 4 * sum(p.numel() for n, p in net.named_parameters() if "weight_v" in n)
    for tile in (4, 16):
        net.sdf_tile_points = tile
        for _ in range(5):
            net.sdf(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(40):
            net.sdf(x)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 40 * 1e3
        print(f"width {width}: weights {nbytes / 1e6:.2f} MB, tile {tile}: {us:7.1f} us per launch -> {nbytes / us / 1e3:6.1f} GB/s per CU")
