"""LaplaceDensity - the no-grad soft clamp applied to the SDF output
(reference: code/model/density_net.py:5-30).  On the fused no-grad path the same expression is
evaluated inside the HIP kernel epilogue (csrc/hm_sdf.hip: sdf_clamp)."""
import torch
import torch.nn as nn


class Density(nn.Module):
    def __init__(self, params_init={}):
        super().__init__()
        for p in params_init:
            setattr(self, p, nn.Parameter(torch.tensor(params_init[p])))

    def forward(self, sdf, beta=None, compute_grad=False):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    """alpha * Laplace(0, beta).cdf(-sdf), alpha = 1/beta, beta = |beta| + beta_min."""

    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.beta_min = torch.tensor(beta_min)
        self._beta_min = float(beta_min)  # added as a scalar: no host->device copy per call (graph capturable)

    @torch.no_grad()
    def density_func(self, sdf, beta=None):
        if beta is None:
            beta = self.get_beta()
        else:
            beta = self.beta.abs() + self._beta_min
        alpha = 1 / beta
        return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))

    def get_beta(self):
        return self.beta.abs() + self._beta_min
