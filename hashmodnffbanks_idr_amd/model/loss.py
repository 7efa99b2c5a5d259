"""IDRLoss = L1(rgb)/N + w_eik * eikonal + w_mask * (1/alpha) * BCE(-alpha*sdf, mask)/N
(reference: code/model/loss.py:4-70).  Scalar reductions over per-ray outputs - elementwise torch."""
import torch
from torch import nn
from torch.nn import functional as F


class IDRLoss(nn.Module):
    def __init__(self, eikonal_weight, mask_weight, alpha):
        super().__init__()
        self.eikonal_weight = eikonal_weight
        self.mask_weight = mask_weight
        self.alpha = alpha
        self.l1_loss = nn.L1Loss(reduction='sum')

    def get_rgb_loss(self, rgb_values, rgb_gt, network_object_mask, object_mask):
        sel = network_object_mask & object_mask
        if sel.sum() == 0:
            return torch.tensor(0.0, device=rgb_values.device).float()
        return self.l1_loss(rgb_values[sel], rgb_gt.reshape(-1, 3)[sel]) / float(object_mask.shape[0])

    def get_eikonal_loss(self, grad_theta):
        if grad_theta.shape[0] == 0:
            return torch.tensor(0.0, device=grad_theta.device).float()
        return ((grad_theta.norm(2, dim=1) - 1) ** 2).mean()

    def get_mask_loss(self, sdf_output, network_object_mask, object_mask):
        mask = ~(network_object_mask & object_mask)
        if mask.sum() == 0:
            return torch.tensor(0.0, device=sdf_output.device).float()
        sdf_pred = -self.alpha * sdf_output[mask]
        gt = object_mask[mask].float()
        # the reference squeezes ALL dims here (loss.py:46) and therefore raises when exactly one ray is
        # in the mask set; reshape(-1) is the same tensor in every other case
        bce = F.binary_cross_entropy_with_logits(sdf_pred.reshape(-1), gt, reduction='sum')
        return (1 / self.alpha) * bce / float(object_mask.shape[0])

    def forward(self, model_outputs, ground_truth):
        rgb_gt = ground_truth['rgb'].to(model_outputs['rgb_values'].device)
        network_object_mask = model_outputs['network_object_mask']
        object_mask = model_outputs['object_mask']
        rgb_loss = self.get_rgb_loss(model_outputs['rgb_values'], rgb_gt, network_object_mask, object_mask)
        mask_loss = self.get_mask_loss(model_outputs['sdf_output'], network_object_mask, object_mask)
        eikonal_loss = self.get_eikonal_loss(model_outputs['grad_theta'])
        loss = rgb_loss + self.eikonal_weight * eikonal_loss + self.mask_weight * mask_loss
        return {'loss': loss, 'rgb_loss': rgb_loss, 'eikonal_loss': eikonal_loss, 'mask_loss': mask_loss}
