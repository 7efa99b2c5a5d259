"""Where along the 100 uniform samples does the sampler find its first sign change?  (bench batch, geometric init,
and after some training steps.)  Samples after the first negative one are never read by the reference algorithm."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import numpy as np
import torch
import bench
from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing

dev = torch.device("cuda", 0)
model = bench._build("C2", dev, 0.0)
inp, gt = bench.synthetic_batch(1234, 2048, dev)
rt = model.ray_tracer
rt.use_device_tracer = False
rec = {}
orig = RayTracing._sample_and_secant


def spy(self, sdf, cams, dirs, t0, t1, true_obj):
    M, n = dirs.shape[0], self.n_steps
    frac = self._linspace(dirs.device).view(1, n)
    ts = t0.unsqueeze(-1) + frac * (t1 - t0).unsqueeze(-1)
    pts = cams.unsqueeze(1) + ts.unsqueeze(-1) * dirs.unsqueeze(1)
    vals = sdf(pts.reshape(-1, 3)).reshape(M, n)
    rank = torch.arange(n, 0, -1, device=dirs.device, dtype=torch.float32).view(1, n)
    first = torch.argmin(torch.sign(vals) * rank, -1)
    hit = vals[torch.arange(M), first] < 0
    rec["first"], rec["hit"], rec["true"] = first.cpu().numpy(), hit.cpu().numpy(), true_obj.cpu().numpy()
    return orig(self, sdf, cams, dirs, t0, t1, true_obj)


RayTracing._sample_and_secant = spy
out = model(inp)
f, h, t = rec["first"], rec["hit"], rec["true"]
need = np.where(h & t, np.where(f == 0, 2, f + 1), 100)
print("sampler rays", len(f), "hits", int(h.sum()), "first-index histogram (hits):", np.bincount(f[h], minlength=100)[:20], "...")
print("samples needed: total", int(need.sum()), "of", 100 * len(f), " mean per ray", need.mean())
for chunk in (4, 8, 16, 25):
    tot = 0
    for fi, hi, ti in zip(f, h, t):
        if hi and ti:
            k = (fi // chunk + 1) * chunk
            tot += min(k, 100) + (1 if fi == 0 and k < 100 else 0)
        else:
            tot += 100
    print(f"chunked early exit, chunk {chunk}: {tot} evaluations")
