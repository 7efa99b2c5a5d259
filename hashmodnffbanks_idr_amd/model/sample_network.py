"""SampleNetwork - first-order re-parametrisation of the ray/surface hit point as a
differentiable function of the network parameters (IDR eq. 3;
reference: code/model/sample_network.py:10-20).  Elementwise, O(N_surface): stays in torch."""
import torch
import torch.nn as nn


class SampleNetwork(nn.Module):
    def forward(self, surface_output, surface_sdf_values, surface_points_grad, surface_dists, surface_cam_loc,
                surface_ray_dirs):
        dirs0 = surface_ray_dirs.detach()
        # <grad sdf(x0), v>  per point
        denom = torch.bmm(surface_points_grad.view(-1, 1, 3), dirs0.view(-1, 3, 1)).squeeze(-1)
        t_theta = surface_dists - (surface_output - surface_sdf_values) / denom
        return surface_cam_loc + t_theta * surface_ray_dirs
