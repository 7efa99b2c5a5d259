import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import numpy as np, torch, bench
from helpers import idr_conf
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
torch.manual_seed(0)
model = IDRNetwork(idr_conf("C1")).cuda()
with torch.no_grad():
    model.implicit_network.lin0.weight_v[:, 3:].normal_(0, 0.02)
    model.implicit_network.embed_model.embedder_obj.table.uniform_(-0.05, 0.05)
model.train()
net = model.implicit_network
x_all = (torch.rand(768, 3, device="cuda") * 2 - 1)
cache = {}
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import _fold_all
fc = {}
net._fold_cache = fc
_fold_all((net,), fc)
out_all, g_all = net.forward_with_gradient(x_all.clone(), cache_out=cache)
x2 = x_all[256:].clone()
out_a, g_a = net.forward_with_gradient(x2.clone())
out_b, g_b = net.forward_with_gradient(x2.clone(), reuse=(cache, 256, 512))
print("out  sep vs slice :", (out_a - out_all[256:]).abs().max().item(), " reuse vs slice:", (out_b - out_all[256:]).abs().max().item())
print("g    sep vs slice :", (g_a - g_all[256:]).abs().max().item(), " reuse vs slice:", (g_b - g_all[256:]).abs().max().item())
for l, (za, zb) in enumerate(zip(cache["z_list"], cache["z_list"])):
    pass
