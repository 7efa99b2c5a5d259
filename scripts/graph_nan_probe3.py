import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch
from helpers import idr_conf
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
FUSED = int(sys.argv[1]); WHAT = sys.argv[2]
torch.manual_seed(0)
model = IDRNetwork(idr_conf("C1")).cuda(); model.train()
net = model.implicit_network; net.use_fused_mlp_grad = bool(FUSED)
x0 = torch.rand(3000, 3, device="cuda") * 2 - 1
def body():
    model.zero_grad(set_to_none=True)
    if WHAT == "nonleaf":
        o1, g1 = net.forward_with_gradient(x0.clone())
        xx = x0 + (o1[:, 0:1] - o1[:, 0:1].detach()) * g1[:, 0, :].detach()
    else:
        xx = x0.clone()
    out, g = net.forward_with_gradient(xx)
    if WHAT == "g":
        loss = (g ** 2).sum()
    elif WHAT == "out":
        loss = out.sum()
    else:
        loss = (g ** 2).sum() + out.sum()
    loss.backward()
    return out, g, loss
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        out, g, loss = body()
torch.cuda.synchronize()
ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
out_ref, g_ref = out.detach().clone(), g.detach().clone()
print("eager loss", float(loss), flush=True)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out, g, loss = body()
gr.replay(); torch.cuda.synchronize()
bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.allclose(p.grad, ref[n], rtol=1e-3, atol=1e-5)]
print("fwd match: out", float((out - out_ref).abs().max()), "g", float((g - g_ref).abs().max()), flush=True)
for n in bad[:40]:
    pg = dict(model.named_parameters())[n].grad
    print("   ", n, "max|diff|", float((pg - ref[n]).abs().max()), "max|ref|", float(ref[n].abs().max()), flush=True)
print("FUSED", FUSED, WHAT, "replay loss", float(loss), "out nan", bool(torch.isnan(out).any()), "g nan", bool(torch.isnan(g).any()),
      "x0 nan", bool(torch.isnan(x0).any()), "mismatching grads:", bad[:8], len(bad), flush=True)
