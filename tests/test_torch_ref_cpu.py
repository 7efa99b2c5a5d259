"""Pins oracle/torch_ref.py (torch-CPU restatement with autograd) against the reference fixtures:
SDF forward, first-order grads, gradient(), eikonal double backward, and one full IDR step."""
import numpy as np
import pytest
import torch

from helpers import make_implicit
from oracle import torch_ref as R


def _ref_implicit(g, cfg):
    net = make_implicit(cfg, tuple(g["hidden"].tolist()), int(g["fvs"]), int(g["seed"]), float(g["perturb"]),
                        float(g["table_scale"]), device="cpu")
    return R.RefImplicit(R._grid_from(net.embed_model.embedder_obj), R._lins(net), net.skip_in)


def _map(name):
    # RefImplicit parameter name -> key used in the fixtures
    if name.startswith("grid.tables."):
        return f"table{name.split('.')[-1]}"
    kind, l = name[0], name[1:]
    return {"v": f"lin{l}.weight_v", "g": f"lin{l}.weight_g", "b": f"lin{l}.bias"}[kind]


def _check(net, g, label):
    for name, p in net.named_parameters():
        if p.grad is None:
            continue
        key = f"{label}:{_map(name)}"
        arr = p.grad.numpy()
        ref_norm = float(g[key + ":norm"])
        assert abs(np.linalg.norm(arr.astype(np.float64)) - ref_norm) <= 1e-5 * ref_norm + 1e-10, key
        if key + ":full" in g.files:
            np.testing.assert_allclose(arr, g[key + ":full"], rtol=1e-4, atol=1e-4 * max(np.abs(g[key + ":full"]).max(), 1e-12))


@pytest.mark.parametrize("tag,cfg", [("narrow", "tiny"), ("full", "C1"), ("C2", "C2")])
def test_sdf_grads(golden, tag, cfg):
    g = golden(f"sdf_{tag}")
    net = _ref_implicit(g, cfg)
    x = torch.from_numpy(g["x"].copy()).requires_grad_(True)
    out = net(x)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-5, atol=2e-6)
    (out * torch.from_numpy(g["R"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx_first"], rtol=1e-5, atol=1e-5 * np.abs(g["dx_first"]).max())
    _check(net, g, "g1")
    net.zero_grad()
    gr = net.gradient(torch.from_numpy(g["x"].copy()))
    np.testing.assert_allclose(gr.detach().numpy()[:, 0], g["gradient"], rtol=1e-5, atol=1e-6)
    eik = ((gr[:, 0, :].norm(2, dim=1) - 1) ** 2).mean()
    assert abs(eik.item() - float(g["eik"])) <= 1e-6
    eik.backward()
    _check(net, g, "g2")


@pytest.mark.parametrize("cfg", ["C1", "C2"])      # C2 = the benchmarked configuration, 2048 rays
def test_idr_step0(golden, cfg):
    from helpers import make_idr
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    g = golden(f"idr_step_{cfg}")
    model = make_idr(cfg, int(g["seed"]), float(g["bias"]) if "bias" in g.files else 0.6, device="cpu")
    ref = R.RefIDR(model)
    ref.train()
    inp = {k: torch.from_numpy(g[k]) for k in ("intrinsics", "uv", "pose", "object_mask")}
    torch.set_num_threads(8)
    torch.manual_seed(1000)
    out = ref(inp)
    lo = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)(out, {"rgb": torch.from_numpy(g["rgb_gt"])})
    assert np.array_equal(out["network_object_mask"].numpy(), g["s0:network_object_mask"])
    for k in ("loss", "rgb_loss", "eikonal_loss", "mask_loss"):
        assert abs(lo[k].item() - float(g[f"s0:{k}"])) <= 1e-5 * abs(float(g[f"s0:{k}"])) + 1e-7, k
    lo["loss"].backward()
    gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=1.0)
    assert abs(gn.item() - float(g["s0:total_grad_norm"])) <= 1e-4 * float(g["s0:total_grad_norm"])
