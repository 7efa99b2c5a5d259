"""Active SDF evaluations per march round of the device tracer (profiling helper): reads the cursor array
out of the tracer workspace after a training-mode forward."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch, bench
from hashmodnffbanks_idr_amd import parallel
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
from hashmodnffbanks_idr_amd.model.loss import IDRLoss
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = IDRNetwork(bench.idr_conf("C2")).to(dev); model.train()
loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
opt = torch.optim.Adam(model.parameters(), lr=1.0e-4)
inp, gt = bench.synthetic_batch(1234, 2048, dev)
torch.manual_seed(100)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    parallel.train_step(model, loss_fn, opt, inp, gt, None)
    if step in (0, 20, 100, 299):
        torch.cuda.synchronize()
        st = model.ray_tracer.last_stats
        v = model.ray_tracer._ws.view(torch.int32).cpu()
        ns = st["sampler_rays"]
        hit = ((v[:-2] == ns) & (v[1:-1] == ns * 100) & (v[2:] == st["secant_rays"])).nonzero().flatten()
        for h in hit.tolist():
            if h >= 64 and int(v[h - 64]) == 4096:
                print("step", step, st, "rounds:", [int(c) for c in v[h - 64:h - 64 + 42]], flush=True)
                break
