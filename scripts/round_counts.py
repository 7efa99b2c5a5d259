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


def counters(ws, n, n_steps=100):
    """the int32 counter block of the tracer workspace (csrc/hm_trace.hip: make_layout)"""
    al = lambda o: (o + 255) // 256 * 256  # noqa: E731
    cap = n * n_steps
    o = al(4 * 13 * n)
    o = al(o + 4 * 6 * n)
    o = al(o + 6 * n)
    o = al(o + 4 * 3 * 3 * cap)
    o = al(o + 4 * 3 * cap)
    return ws.view(torch.uint8)[o:o + 4 * 80].view(torch.int32).cpu()


for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    parallel.train_step(model, loss_fn, opt, inp, gt, None)
    if step in (0, 20, 100, 200, 299):
        torch.cuda.synchronize()
        st = model.ray_tracer.last_stats
        c = counters(model.ray_tracer._ws, 2048)
        print("step", step, st, "rounds:", [int(v) for v in c[:42]], flush=True)
