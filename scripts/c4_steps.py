"""A few captured training iterations at BASELINE configs[3]'s table size (T = 2^22: 223.5 MiB table, 2048 rays per GPU)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch, bench
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
from hashmodnffbanks_idr_amd.model.loss import IDRLoss
from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
from hashmodnffbanks_idr_amd.training.optim import ClipAdam
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = IDRNetwork(bench.idr_conf(cfg)).to(dev); model.train()
loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
st = GraphedTrainStep(model, loss_fn, ClipAdam(model.parameters(), lr=1e-4), warmup=2)
inp, gt = bench.synthetic_batch(1234, 2048, dev)
torch.manual_seed(100)
for i in range(5):
    out, lo = st.step(inp, gt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20):
    out, lo = st.step(inp, gt)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(cfg, "ms/step", round(dt * 1e3, 3), "loss", float(lo["loss"]), model.ray_tracer.last_stats, "graph", st.g_fb is not None)
