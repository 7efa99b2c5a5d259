"""SIREN activation and its initialisers (reference: code/model/embeddings/Sine.py)."""
import numpy as np
import torch
from torch import nn


class Sine(nn.Module):
    """y = sin(w0 * x)"""

    def __init__(self, w0):
        super().__init__()
        self.w0 = w0

    def forward(self, input, compute_grad=False):
        return torch.sin(input * self.w0)


def sine_init(m, w0, num_input=None):
    """hidden SIREN layer: U(+-sqrt(6/fan_in)/w0) for weight AND bias (Sine.py:14-19)"""
    if hasattr(m, 'weight') and num_input is None:
        bound = np.sqrt(6 / m.weight.size(-1)) / w0
        torch.nn.init.uniform_(m.weight, -bound, bound)
        torch.nn.init.uniform_(m.bias, -bound, bound)


def first_layer_sine_init(m):
    """first SIREN layer: U(+-1/fan_in) (Sine.py:21-25)"""
    if hasattr(m, 'weight'):
        bound = 1.0 / m.weight.size(-1)
        torch.nn.init.uniform_(m.weight, -bound, bound)
        torch.nn.init.uniform_(m.bias, -bound, bound)
