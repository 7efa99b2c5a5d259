"""micro-benchmark of the small-tile SDF bodies: time per launch for n live points x tile size (4 / 8 / 16 points per
workgroup).  Question: is a sparse round bound by each CU's own weight stream (then fewer, fuller workgroups cost the
same) or by the aggregate L2 rate of 256 CUs streaming 7.9 MB each (then 128 workgroups of 8 points beat 256 of 4)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import numpy as np, torch, bench
model = bench._build("C2", torch.device("cuda", 0), 0.0)
net = model.implicit_network
g = torch.Generator(device="cpu").manual_seed(1)
for n in (256, 512, 1000, 1024, 1500, 2048, 3000, 4096):
    x = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
    row = []
    for tile in (4, 8, 16):
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
        row.append(s.elapsed_time(e) / 40 * 1e3)
    print(f"n={n:5d}  tile4 {row[0]:7.1f} us ({(n+3)//4:4d} wg)   tile8 {row[1]:7.1f} us ({(n+7)//8:4d} wg)   tile16 {row[2]:7.1f} us ({(n+15)//16:4d} wg)")
