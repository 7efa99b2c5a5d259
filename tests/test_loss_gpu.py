"""hm_idr_loss (one-launch IDRLoss value + gradients) against the torch formulation of code/model/loss.py:4-70."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,case", [(2048, 3072, "mixed"), (1, 1, "mixed"), (777, 0, "mixed"), (300, 450, "all_surface"),
                                     (300, 450, "no_surface")])
def test_fused_loss_matches_torch(n, m, case):
    from hashmodnffbanks_idr_amd.model import loss as L
    g = torch.Generator(device="cpu").manual_seed(n + m)
    rgb = torch.rand(n, 3, generator=g).cuda().requires_grad_(True)
    gt = torch.rand(1, n, 3, generator=g).cuda()
    sdf = (torch.randn(n, 1, generator=g) * 0.05).cuda().requires_grad_(True)
    hit = torch.rand(n, generator=g) > 0.4
    inside = torch.rand(n, generator=g) > 0.3
    if case == "all_surface":
        hit[:], inside[:] = True, True
    if case == "no_surface":
        hit[:] = False
    grad = torch.randn(m, 3, generator=g).cuda()
    if m > 2:
        grad[1] = 0.0                       # zero vector: norm backward is defined as 0
    grad.requires_grad_(True)
    out = {"rgb_values": rgb, "sdf_output": sdf, "grad_theta": grad, "network_object_mask": hit.cuda(),
           "object_mask": inside.cuda()}
    ours = L.idr_loss_terms(out, gt, 0.1, 100.0, 50.0)
    (ours["loss"] * 1.7).backward()
    got = [t.grad.clone() if t.grad is not None else torch.zeros_like(t) for t in (rgb, sdf, grad)]
    for t in (rgb, sdf, grad):
        t.grad = None
    ref = L.idr_loss_terms_torch(out, gt, 0.1, 100.0, 50.0)
    (ref["loss"] * 1.7).backward()
    for k in ("loss", "rgb_loss", "eikonal_loss", "mask_loss"):
        np.testing.assert_allclose(float(ours[k].detach()), float(ref[k].detach()), rtol=3e-6, atol=1e-7, err_msg=k)
    for a, t, name in zip(got, (rgb, sdf, grad), ("d_rgb", "d_sdf", "d_grad")):
        refg = t.grad if t.grad is not None else torch.zeros_like(t)
        np.testing.assert_allclose(a.cpu().numpy(), refg.cpu().numpy(), rtol=2e-5, atol=1e-9, err_msg=name)
