"""Frequency encoders with the reference's names and constructor arguments
(reference: code/model/embeddings/frequency_enc.py).

FourierFeature is part of the MultiResHashGridMLP output.  On the no-grad / table-only-grad
path it is computed inside the fused HIP encoder kernel (csrc/hm_encode.hip); the torch
expression below is the any-order-differentiable form used when the INPUT POINTS need
gradients (ImplicitNetwork.gradient, create_graph=True).
"""
import numpy as np
import torch
import torch.nn as nn


class FourierFeature(nn.Module):
    """[x, sin(2*pi*x@B), cos(2*pi*x@B)] with a persistent random buffer B [in,channels]
    (reference frequency_enc.py:54-67)."""

    def __init__(self, input_dims=3, sigma=1.0, num_channels=256, include_input=True):
        super().__init__()
        self.input_dims = input_dims
        self.include_input = include_input
        self.register_buffer('B', torch.randn(input_dims, int(num_channels)) * sigma, persistent=True)
        self.embeddings_dim = 2 * num_channels + 3 if include_input else 2 * num_channels

    def forward(self, x, compute_grad=False):
        xp = torch.matmul(2 * np.pi * x, self.B.to(x.device))
        ff = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
        return torch.cat([x, ff], dim=-1) if self.include_input else ff


class PositionalEncoding(nn.Module):
    """NeRF positional encoding (reference frequency_enc.py:6-51); used per level inside NFFB
    and by get_embedder for view directions."""

    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        self.include_input = kwargs['include_input']
        d = kwargs['input_dims']
        max_freq = kwargs['max_freq_log2']
        n_freqs = kwargs['num_freqs']
        if kwargs['log_sampling']:
            self.freq_bands = 2. ** torch.linspace(0., max_freq, n_freqs)
        else:
            self.freq_bands = torch.linspace(2. ** 0., 2. ** max_freq, n_freqs)
        self.periodic_fns = kwargs['periodic_fns']
        # the reference counts the input once in out_dim and once more in embeddings_dim
        self.out_dim = d + d * len(self.freq_bands) * len(self.periodic_fns)
        self.embeddings_dim = self.out_dim + d if self.include_input else self.out_dim

    def embed(self, inputs):
        if (self.include_input and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 2
                and inputs.requires_grad and torch.is_grad_enabled() and len(self.freq_bands) <= 16
                and list(self.periodic_fns) == [torch.sin, torch.cos]):
            # grad path of the filter-bank embedders: the whole row and its first / second order passes are one kernel
            # each (csrc/hm_elem.hip) instead of ~26 / ~40 / ~60 elementwise launches per chunk
            from ... import ops
            return ops.posenc(inputs, self.freq_bands.tolist())
        parts = [inputs] if self.include_input else []
        for freq in self.freq_bands:
            for fn in self.periodic_fns:
                parts.append(fn(inputs * freq))
        enc = torch.cat(parts, -1)
        return torch.cat([inputs, enc], -1) if self.include_input else enc

    def forward(self, inputs, compute_grad=False):
        return self.embed(inputs)


def get_embedder(multires):
    """reference frequency_enc.py:154-169"""
    eo = PositionalEncoding(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                            log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim
