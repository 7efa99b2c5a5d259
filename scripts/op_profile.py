"""Operator-level attribution of one eager static-shape training step (profiling helper)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch, bench
from torch.profiler import profile, ProfilerActivity
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
from hashmodnffbanks_idr_amd.model.loss import IDRLoss
from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
torch.manual_seed(0)
model = IDRNetwork(bench.idr_conf("C2")).cuda(); model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
st = GraphedTrainStep(model, IDRLoss(0.1, 100.0, 50.0), opt, use_graph=False)
inp, gt = bench.synthetic_batch(1234, 2048, "cuda")
for _ in range(8):
    st.step(inp, gt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        st.step(inp, gt)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
