import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import torch, bench
from helpers import idr_conf
from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
from hashmodnffbanks_idr_amd.model.loss import idr_loss_terms
mode = sys.argv[1]; FUSED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
model = IDRNetwork(idr_conf("C1")).cuda(); model.train(); model.implicit_network.use_fused_mlp_grad = bool(FUSED)
inp, gt = bench.synthetic_batch(1, 512, "cuda")
eik = torch.rand(256, 3, device="cuda") * 2 - 1
steps = torch.rand(100, device="cuda")
def body():
    model.zero_grad(set_to_none=True)
    out = model.forward_static(inp, eik, steps)
    lo = idr_loss_terms(out, gt["rgb"], 0.1, 100.0, 50.0)
    if mode == "bwd":
        lo["loss"].backward()
    return out, lo
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        out, lo = body()
torch.cuda.synchronize()
print("eager loss", float(lo["loss"]), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out, lo = body()
for r in range(2):
    g.replay(); torch.cuda.synchronize()
    print(mode, "replay", r, "loss", float(lo["loss"]), {k: bool(torch.isnan(v).any()) for k, v in out.items() if torch.is_tensor(v) and v.dtype.is_floating_point},
          "nan grads", sum(int(torch.isnan(p.grad).any()) for p in model.parameters() if p.grad is not None), flush=True)
print("params nan:", [n for n, p in model.named_parameters() if torch.isnan(p).any()][:5], "inp nan:", {k: bool(torch.isnan(v).any()) for k, v in inp.items() if v.dtype.is_floating_point},
      "eik", bool(torch.isnan(eik).any()), "steps", bool(torch.isnan(steps).any()), "B", bool(torch.isnan(model.implicit_network.embed_model.embedder_obj.freq_encoding.B).any()), flush=True)
pk = model.implicit_network._packed
print("packed nan", [bool(torch.isnan(b[0]).any() or torch.isnan(b[2]).any()) for b in pk.bufs], flush=True)
with torch.no_grad():
    x = torch.rand(1000, 3, device="cuda") * 2 - 1
    print("eager sdf after replay nan:", bool(torch.isnan(model.implicit_network.sdf(x)).any()), flush=True)
    from hashmodnffbanks_idr_amd.utils import rend_util
    rd, cl = rend_util.get_camera_params(inp["uv"], inp["pose"], inp["intrinsics"])
    model.ray_tracer.steps_override = steps
    p, m, d = model.ray_tracer(sdf=model.implicit_network.sdf, cam_loc=cl, object_mask=inp["object_mask"].reshape(-1), ray_directions=rd)
    print("eager tracer after replay: dists nan", bool(torch.isnan(d).any()), "stats", model.ray_tracer.last_stats, flush=True)
