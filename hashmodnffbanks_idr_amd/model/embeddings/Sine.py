"""SIREN pieces used by the Fourier-filter-bank trunk (reference: code/model/embeddings/Sine.py):
the activation sin(w0 * x) and the two uniform initialisers (weights AND biases drawn, in that order,
so the RNG stream matches the reference constructor)."""
import math

import torch
from torch import nn


class Sine(nn.Module):
    def __init__(self, w0):
        super().__init__()
        self.w0 = w0

    def forward(self, input, compute_grad=False):
        if input.is_cuda and input.dtype == torch.float32 and input.requires_grad and torch.is_grad_enabled():
            # grad path of the filter-bank trunk: one kernel per pass (forward, backward, double backward) instead
            # of autograd's chain of mul / sin / cos / neg kernels
            from ... import ops
            return ops.sine(input, self.w0)
        return torch.sin(input * self.w0)


def _uniform_both(layer, bound):
    torch.nn.init.uniform_(layer.weight, -bound, bound)
    torch.nn.init.uniform_(layer.bias, -bound, bound)


def sine_init(m, w0, num_input=None):
    """hidden layers: U(+-sqrt(6 / fan_in) / w0)   (only when fan-in is left implicit, as the reference does)"""
    if hasattr(m, 'weight') and num_input is None:
        _uniform_both(m, math.sqrt(6 / m.weight.size(-1)) / w0)


def first_layer_sine_init(m):
    """first layer: U(+-1 / fan_in)"""
    if hasattr(m, 'weight'):
        _uniform_both(m, 1.0 / m.weight.size(-1))
