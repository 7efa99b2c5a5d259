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

    def forward(self, content, style):
        content_features = content.view(-1, self.d_in)
        style_features = style.view(-1, self.feature_vector_size)
        modulated = ops.linear(style_features, self.linear_transform.weight, self.linear_transform.bias)
        attention_weights = F.softmax(ops.linear(content_features, self.attention.weight, self.attention.bias), dim=1)
        weighted = attention_weights * modulated
        mean = weighted.mean(dim=1, keepdim=True)
        var = weighted.var(dim=1, unbiased=False, keepdim=True)
        return (weighted - mean) / torch.sqrt(var + self.eps)
