import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch, bench
from helpers import idr_conf
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
from hashmodnffbanks_idr_amd.model.loss import IDRLoss
from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
fused = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
CFG = sys.argv[2] if len(sys.argv) > 2 else "C1"; NR = int(sys.argv[3]) if len(sys.argv) > 3 else 512
model = IDRNetwork(bench.idr_conf(CFG) if CFG == "C2" else idr_conf(CFG)).cuda(); model.train()
model.implicit_network.use_fused_mlp_grad = bool(fused)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
st = GraphedTrainStep(model, IDRLoss(0.1, 100.0, 50.0), opt, warmup=2)
inp, gt = bench.synthetic_batch(1, NR, "cuda")
for i in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3):
    out, lo = st.step(inp, gt)
    if os.environ.get("NOSYNC") and i < int(sys.argv[4]) - 1:
        continue
    torch.cuda.synchronize()
    bad = {k: bool(torch.isnan(v).any()) for k, v in out.items() if torch.is_tensor(v) and v.dtype.is_floating_point}
    gbad = [n for n, p in model.named_parameters() if p.grad is not None and torch.isnan(p.grad).any()]
    pbad = [n for n, p in model.named_parameters() if torch.isnan(p).any()]
    if i == 2:
        pk = model.implicit_network._packed
        print("after first replay: packed NaN", [bool(torch.isnan(b[0]).any()) for b in pk.bufs], "beta", pk.desc.beta, flush=True)
    print("step", i, float(lo["loss"]), bad, "nan grads:", gbad[:6], len(gbad), "nan params:", len(pbad), flush=True)
pk = model.implicit_network._packed
print("packed NaN per layer:", [bool(torch.isnan(b[0]).any() or torch.isnan(b[1]).any() or torch.isnan(b[2]).any()) for b in pk.bufs], flush=True)
