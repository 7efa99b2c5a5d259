"""LaplaceDensity: rho(s) = (1/beta) * (1/2 + 1/2 * sign(s) * expm1(-|s| / beta)),  beta = |beta_param| + beta_min,
evaluated WITHOUT gradient and used by ImplicitNetwork to soft-clamp its SDF column,
sdf = tanh(s / (2 + rho(s)))   (reference: code/model/density_net.py:16-30,
code/model/implicit_differentiable_renderer.py:112).

On the fused no-grad path the same expression runs in the HIP kernel epilogue
(csrc/hm_sdf.hip: sdf_clamp); this module is the grad-enabled path's version and the owner of the
`beta` parameter (state_dict key `dencity_net.beta`; it never receives a gradient, exactly as in the
reference, so Adam and clip_grad_norm_ skip it).
"""
import torch
import torch.nn as nn


class LaplaceDensity(nn.Module):
    def __init__(self, params_init=None, beta_min=0.0001):
        super().__init__()
        for name, value in (params_init or {}).items():
            self.register_parameter(name, nn.Parameter(torch.tensor(value)))
        self._beta_min = float(beta_min)   # added as a scalar: no host->device copy per call (graph capturable)

    def get_beta(self):
        return self.beta.abs() + self._beta_min

    @torch.no_grad()
    def density_func(self, sdf, beta=None):
        b = self.get_beta()
        return (1 / b) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / b))

    def forward(self, sdf, beta=None, compute_grad=False):
        return self.density_func(sdf, beta=beta)
