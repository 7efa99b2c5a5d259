"""training.optim.ClipAdam (csrc/hm_optim.hip) against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam
(the reference iteration tail, training/idr_train.py:128,306-309)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(5217, 2), (512, 67), (512,), (1,), (3, 16), (257, 512), (100003,), (7, 9, 5)]


def _make(seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.randn(s, generator=g).to(dev) for s in SHAPES]


@pytest.mark.parametrize("max_norm,gscale", [(1.0, 3.0), (1.0, 1e-3), (None, 1.0)])
def test_clip_adam_matches_torch(max_norm, gscale):
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    dev = torch.device("cuda")
    pa = [torch.nn.Parameter(t.clone()) for t in _make(0, dev)]
    pb = [torch.nn.Parameter(t.clone()) for t in _make(0, dev)]
    ours = ClipAdam(pa, lr=1e-3, max_norm=max_norm)
    ref = torch.optim.Adam(pb, lr=1e-3)
    for it in range(6):
        grads = _make(100 + it, dev)
        for p, q, g in zip(pa, pb, grads):
            p.grad = (g * gscale).clone()
            q.grad = (g * gscale).clone()
        if it == 3:                      # a parameter without gradient is skipped by both
            pa[2].grad = None
            pb[2].grad = None
        if max_norm is not None:
            total = torch.nn.utils.clip_grad_norm_(pb, max_norm=max_norm)
        ref.step()
        ours.step()
        if max_norm is not None:
            np.testing.assert_allclose(float(ours.last_grad_norm), float(total), rtol=2e-6)
        for p, q in zip(pa, pb):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
            if p.grad is not None:       # gradients are clipped in place, like clip_grad_norm_
                np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.cpu().numpy(), rtol=2e-6, atol=1e-9)
    # optimizer checkpoints are interchangeable with torch.optim.Adam's
    sd = ours.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][0]["step"]) == 6.0
    ref2 = torch.optim.Adam([torch.nn.Parameter(t.clone()) for t in _make(0, dev)], lr=1e-3)
    ref2.load_state_dict(copy.deepcopy(sd))
    ours2 = ClipAdam([torch.nn.Parameter(p.detach().clone()) for p in pb], lr=1e-3, max_norm=max_norm)
    ours2.load_state_dict(copy.deepcopy(ref.state_dict()))   # (load_state_dict aliases tensors otherwise)
    for p, q, g in zip(ours2.param_groups[0]["params"], pb, _make(999, dev)):
        p.grad = g.clone()
        q.grad = g.clone()
    if max_norm is not None:
        torch.nn.utils.clip_grad_norm_(pb, max_norm=max_norm)
    ref.step()
    ours2.step()
    for p, q in zip(ours2.param_groups[0]["params"], pb):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


def test_clip_adam_many_tensors():
    """more tensors than one launch's by-value table holds (HM_ADAM_MAX_TENSORS = 80)"""
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    base = [torch.randn(11 + i, generator=g) for i in range(200)]
    pa = [torch.nn.Parameter(t.clone().to(dev)) for t in base]
    pb = [torch.nn.Parameter(t.clone().to(dev)) for t in base]
    ours, ref = ClipAdam(pa, lr=1e-2, max_norm=0.5), torch.optim.Adam(pb, lr=1e-2)
    for it in range(3):
        for p, q in zip(pa, pb):
            gr = torch.randn(p.shape, generator=g).to(dev)
            p.grad, q.grad = gr.clone(), gr.clone()
        torch.nn.utils.clip_grad_norm_(pb, max_norm=0.5)
        ref.step()
        ours.step()
    for p, q in zip(pa, pb):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
