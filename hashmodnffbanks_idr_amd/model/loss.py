"""IDRLoss (reference: code/model/loss.py:4-70):

    loss = L1(rgb on surface rays)/N  +  w_eik * mean((|grad sdf| - 1)^2)
         + w_mask * (1/alpha) * BCE(-alpha * sdf, object_mask on the other rays)/N

Formulated with ray masks instead of the reference's boolean gathers: the selected-subset sums become
masked sums over all N rays.  Same terms and values (an empty selection contributes an exact 0, which is
what the reference's `.sum() == 0` guards return), but no data-dependent shapes and no device->host
reads, so the loss can sit inside a captured HIP graph (training/graph_step.py).
"""
import torch
from torch import nn
from torch.nn import functional as F


class _IdrLossHip(torch.autograd.Function):
    """terms = [loss, rgb_loss, eikonal_loss, mask_loss] from one HIP launch (csrc/hm_loss.hip) that also leaves
    d loss / d (rgb_values, sdf_output, grad_theta); only terms[0] carries gradient."""

    @staticmethod
    def forward(ctx, rgb, sdf, grad_theta, rgb_gt, hit, inside, w_eik, w_mask, alpha):
        from .. import _lib
        rgb, sdf_c, rgb_gt = rgb.contiguous(), sdf.reshape(-1).contiguous(), rgb_gt.reshape(-1, 3).contiguous()
        n = rgb.shape[0]
        m = 0 if grad_theta is None else grad_theta.shape[0]
        gt_c = grad_theta.contiguous() if m else None
        hit8, in8 = hit.reshape(-1).contiguous().view(torch.uint8), inside.reshape(-1).contiguous().view(torch.uint8)
        terms = torch.empty(4, dtype=torch.float32, device=rgb.device)
        d_rgb, d_sdf = torch.empty_like(rgb), torch.empty_like(sdf_c)
        d_grad = torch.empty_like(gt_c) if m else None
        _lib.check(_lib.lib().hm_idr_loss(_lib.dptr(rgb), _lib.dptr(rgb_gt), _lib.dptr(sdf_c), _lib.dptr(hit8),
                                          _lib.dptr(in8), n, _lib.dptr(gt_c), m, float(w_eik), float(w_mask),
                                          float(alpha), _lib.dptr(terms), _lib.dptr(d_rgb), _lib.dptr(d_sdf),
                                          _lib.dptr(d_grad), _lib.stream_ptr(rgb)))
        ctx.save_for_backward(d_rgb, d_sdf, d_grad)
        ctx.sdf_shape = sdf.shape
        # the loss leaves as its OWN tensor: indexing terms[0] outside would put a SelectBackward node in front of this
        # one, whose zeros + copy_ of one float is a MEMCPY node in a captured graph (kernel copy here)
        from .. import ops
        loss = ops.dcopy_(torch.empty(1, dtype=torch.float32, device=rgb.device), terms[0:1])
        ctx.mark_non_differentiable(terms)
        return loss.reshape(()), terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_loss, _d_terms):
        d_rgb, d_sdf, d_grad = ctx.saved_tensors
        g = d_loss
        return (d_rgb * g, (d_sdf * g).reshape(ctx.sdf_shape), d_grad * g if d_grad is not None else None,
                None, None, None, None, None, None)


def idr_loss_terms(model_outputs, rgb_gt, eikonal_weight, mask_weight, alpha):
    if model_outputs['rgb_values'].is_cuda:
        gt = model_outputs['grad_theta']
        loss, terms = _IdrLossHip.apply(model_outputs['rgb_values'], model_outputs['sdf_output'],
                                  gt if (gt is not None and gt.shape[0] > 0) else None, rgb_gt,
                                  model_outputs['network_object_mask'], model_outputs['object_mask'],
                                  eikonal_weight, mask_weight, alpha)
        det = terms.detach()
        return {'loss': loss, 'rgb_loss': det[1], 'eikonal_loss': det[2], 'mask_loss': det[3]}
    return idr_loss_terms_torch(model_outputs, rgb_gt, eikonal_weight, mask_weight, alpha)


def idr_loss_terms_torch(model_outputs, rgb_gt, eikonal_weight, mask_weight, alpha):
    """The same terms composed from torch ops (host-side tensors: the oracle / CPU baseline leg; also the
    checker of the HIP kernel in tests)."""
    hit = model_outputs['network_object_mask']
    inside = model_outputs['object_mask']
    n_rays = float(inside.shape[0])
    surface = hit & inside                      # rays rendered by the network AND inside the object mask
    others = ~surface                           # rays that feed the mask term

    rgb_err = torch.abs(model_outputs['rgb_values'] - rgb_gt.reshape(-1, 3))
    rgb_loss = (rgb_err * surface.unsqueeze(-1)).sum() / n_rays

    grad_theta = model_outputs['grad_theta']
    if grad_theta is None or grad_theta.shape[0] == 0:
        eikonal_loss = torch.zeros((), device=rgb_err.device)
    else:
        eikonal_loss = ((grad_theta.norm(2, dim=1) - 1) ** 2).mean()

    logits = (-alpha * model_outputs['sdf_output']).reshape(-1)
    bce = F.binary_cross_entropy_with_logits(logits, inside.float(), reduction='none')
    mask_loss = (1 / alpha) * (bce * others).sum() / n_rays

    total = rgb_loss + eikonal_weight * eikonal_loss + mask_weight * mask_loss
    return {'loss': total, 'rgb_loss': rgb_loss, 'eikonal_loss': eikonal_loss, 'mask_loss': mask_loss}


class IDRLoss(nn.Module):
    """Same constructor and call contract as the reference class."""

    def __init__(self, eikonal_weight, mask_weight, alpha):
        super().__init__()
        self.eikonal_weight = eikonal_weight
        self.mask_weight = mask_weight
        self.alpha = alpha      # doubled by the runner at alpha_milestones (training/idr_train.py:227-228)

    def forward(self, model_outputs, ground_truth):
        rgb_gt = ground_truth['rgb'].to(model_outputs['rgb_values'].device)
        return idr_loss_terms(model_outputs, rgb_gt, self.eikonal_weight, self.mask_weight, self.alpha)
