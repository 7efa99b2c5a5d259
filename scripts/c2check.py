import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, R+'/tests', R+'/tests/golden']
import torch, numpy as np, bench, params as P
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
torch.manual_seed(0)
m = IDRNetwork(bench.idr_conf("C2")).cuda()
net = m.implicit_network
x = torch.from_numpy(P.make_points(1, 3000, -1, 1)).cuda()
with torch.no_grad():
    for tile in (16, 64):
        net.sdf_tile_points = tile
        f = net(x)
        s = net.sdf(x)
        print(tile, "fused sdf range", s.min().item(), s.max().item(), "full-vs-sdfonly", (f[:,0]-s).abs().max().item())
net.sdf_tile_points = 0
xg = x.clone().requires_grad_(True)
g = net(xg)
print("grad-path vs fused max diff", (g.detach()-f).abs().max().item(), "sdf col", (g[:,0].detach()-s).abs().max().item())
r = x.norm(dim=1)
print("corr: sdf at r<0.3", s[r<0.3].mean().item(), " r>0.9", s[r>0.9].mean().item())
