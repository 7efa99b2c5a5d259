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


def idr_loss_terms(model_outputs, rgb_gt, eikonal_weight, mask_weight, alpha):
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
