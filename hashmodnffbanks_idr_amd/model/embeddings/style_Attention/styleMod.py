"""StyleAttention (reference: code/model/embeddings/style_Attention/styleMod.py:16-43).

What the reference block computes: Linear(style) scaled by softmax(Linear(content), dim=1) - a
softmax over a size-1 dimension, i.e. identically 1 - followed by nn.InstanceNorm1d applied to a 2-D
tensor, i.e. a per-ROW normalisation over the feature dimension (biased variance, eps 1e-5, no
affine).  The attention Linear is kept as a parameter container (state_dict / RNG parity) and still
multiplied in so its (zero) gradient exists exactly as in the reference."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class StyleAttention(nn.Module):
    def __init__(self, d_in=3, feature_vector_size=28):
        super().__init__()
        self.d_in = d_in
        self.feature_vector_size = feature_vector_size
        self.linear_transform = nn.Linear(feature_vector_size, feature_vector_size).to(device=device)
        self.attention = nn.Linear(d_in, 1).to(device=device)
        self.eps = 1e-5
        self.fused_norm = True   # False: the torch expression of the block (softmax, mean / var / sqrt / div)

    def forward(self, content, style):
        content_features = content.view(-1, self.d_in)
        style_features = style.view(-1, self.feature_vector_size)
        modulated = ops.linear(style_features, self.linear_transform.weight, self.linear_transform.bias)
        if (modulated.is_cuda and modulated.requires_grad and torch.is_grad_enabled() and self.fused_norm
                and modulated.shape[1] <= 128):
            # grad path: the softmax over a size-1 dimension is exactly 1.0 and its backward exactly 0, so `weighted`
            # IS `modulated` and the attention Linear (and `content`) receive exact zeros - produced here by a zero-weight
            # term instead of the softmax / K=3 linear chain and its double backward; the row normalisation and its
            # first / second order passes are one kernel each (ops.rownorm) instead of ~10 / ~25 / ~60 launches
            zero = (self.attention.weight.sum() + self.attention.bias.sum()) * 0.0
            return ops.rownorm(modulated, self.eps) + zero
        attention_weights = F.softmax(ops.linear(content_features, self.attention.weight, self.attention.bias), dim=1)
        weighted = attention_weights * modulated
        mean = weighted.mean(dim=1, keepdim=True)
        var = weighted.var(dim=1, unbiased=False, keepdim=True)
        return (weighted - mean) / torch.sqrt(var + self.eps)
