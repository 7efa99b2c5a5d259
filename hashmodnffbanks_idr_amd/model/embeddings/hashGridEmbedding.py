"""MultiResHashGridMLP - same constructor, attributes, output layout and state_dict keys as the
reference (code/model/embeddings/hashGridEmbedding.py:106-155), computed by the HIP kernels
of csrc/hm_encode.hip through the C ABI.

Differences in mechanism (not behaviour):
  * the L per-level nn.Embedding tables live in ONE fused parameter ``table`` [sum(rows), F]
    (one Adam kernel, one all-reduce bucket); ``state_dict()`` still emits / accepts the
    reference keys ``levels.{l}.embedding.weight`` and ``freq_encoding.B``;
  * ``frac_mode="reference"`` (default) reproduces the reference's degenerate interpolation
    weights (xf = x - x.float() == 0, hashGridEmbedding.py:86); ``"trilinear"`` is an opt-in,
    non-parity mode with real interpolation weights, differentiable w.r.t. x to second order
    (csrc/hm_encode_dx.hip), so the eikonal / normal terms train it.
"""
import math

import torch
import torch.nn as nn

from ... import ops
from .frequency_enc import FourierFeature as FrequencyEncoding

HASH_PRIMES = [1, 3, 2654435761]  # the three primes the 3-D path uses (hashGridEmbedding.py:14)
DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class _LevelView:
    """Read-only stand-in for the reference's ``levels[l]`` (an _HashGridMLP with .embedding.weight)."""

    class _Emb:
        def __init__(self, w):
            self.weight = w

    def __init__(self, weight, resolution, hashmap_size, n_features):
        self.embedding = _LevelView._Emb(weight)
        self.resolution = resolution
        self.hashmap_size = hashmap_size
        self.n_features = n_features
        self.dim = 3


class MultiResHashGridMLP(nn.Module):
    def __init__(self, include_input: bool, in_dim: int, n_levels: int, max_points_per_level: int,
                 log2_hashmap_size: int, base_resolution: int, desired_resolution: int,
                 frac_mode: str = "reference"):
        super().__init__()
        if in_dim != 3:
            raise ValueError("MultiResHashGridMLP (HIP): only 3-D inputs are supported")
        if frac_mode not in ops.FRAC_MODES:
            raise ValueError(f"frac_mode must be one of {list(ops.FRAC_MODES)}")
        if not include_input:
            # the reference raises AttributeError here (self.output_dim is never set, :147)
            raise AttributeError("'MultiResHashGridMLP' object has no attribute 'output_dim'")
        self.include_input = include_input
        self.frac_mode = frac_mode
        self.n_levels = n_levels
        self.n_features = max_points_per_level
        res, rows = ops.level_table(n_levels, log2_hashmap_size, base_resolution, desired_resolution, in_dim)
        self.resolutions, self.hashmap_sizes = res, rows
        self.hashmap_size = rows[-1]
        self.desc = ops.GridDesc(res, rows, max_points_per_level)
        # --- initialisation: same RNG consumption, in the same order, as the reference constructor
        # (per level: nn.Embedding's N(0,1) init, then U(-1e-4,1e-4); a fresh FourierFeature per loop turn)
        std = 1e-4
        sigma = (math.log(desired_resolution) - math.log(base_resolution)) / (base_resolution - 1)
        chunks = []
        for l in range(n_levels):
            w = torch.empty(rows[l], max_points_per_level)
            nn.init.normal_(w)
            nn.init.uniform_(w, -std, std)
            chunks.append(w)
            self.freq_encoding = FrequencyEncoding(in_dim, sigma, num_channels=n_levels, include_input=True)
        self.table = nn.Parameter(torch.cat(chunks, 0).to(DEVICE))
        self.freq_encoding = self.freq_encoding.to(DEVICE)
        self.embeddings_dim = in_dim + n_levels * max_points_per_level + (self.freq_encoding.embeddings_dim - in_dim)
        self.grad_collector = None   # parallel.TouchedRowExchange when the table gradient is exchanged sparsely
        self.fused_input_grad = True  # False: d/dx of the Fourier columns through torch's elementwise autograd
        self._register_state_dict_hook(MultiResHashGridMLP._split_table_hook)
        self._register_load_state_dict_pre_hook(self._fuse_table_hook)

    # ---- reference-compatible views -------------------------------------------------------
    @property
    def levels(self):
        off = self.desc.row_off
        return [_LevelView(self.table[int(off[l]):int(off[l + 1])], self.resolutions[l], self.hashmap_sizes[l],
                           self.n_features) for l in range(self.n_levels)]

    @staticmethod
    def _split_table_hook(module, state_dict, prefix, local_metadata):
        t = state_dict.pop(prefix + "table")
        off = module.desc.row_off
        for l in range(module.n_levels):
            state_dict[f"{prefix}levels.{l}.embedding.weight"] = t[int(off[l]):int(off[l + 1])]
        return state_dict

    def _fuse_table_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        keys = [f"{prefix}levels.{l}.embedding.weight" for l in range(self.n_levels)]
        if all(k in state_dict for k in keys):
            parts = [state_dict.pop(k) for k in keys]
            for l, p in enumerate(parts):
                if tuple(p.shape) != (self.hashmap_sizes[l], self.n_features):
                    error_msgs.append(f"size mismatch for {keys[l]}: {tuple(p.shape)} vs "
                                      f"{(self.hashmap_sizes[l], self.n_features)}")
                    return
            state_dict[prefix + "table"] = torch.cat([p.to(self.table.device) for p in parts], 0)

    # ---- forward --------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, compute_grad=False) -> torch.Tensor:
        """x [..., 3] fp32 -> [..., E]: cat[FourierFeature(x), level_0(x) ... level_{L-1}(x)]."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, 3)
        fm = ops.FRAC_MODES[self.frac_mode]
        B = self.freq_encoding.B
        if not torch.is_grad_enabled() or not (x2.requires_grad or self.table.requires_grad):
            out = ops.encode_fwd(self.desc, x2, self.table, B, fm)
        elif not x2.requires_grad:
            # only the table needs a gradient: one fused kernel, backward = table scatter
            out = ops.encode_table_grad(x2, self.table, B, self.desc, fm, self.grad_collector)
        elif self.frac_mode == "reference" and x2.is_cuda and self.fused_input_grad:
            # the points need a gradient too (ImplicitNetwork.gradient, create_graph=True): still ONE encoder launch;
            # the row's dependence on x - the pass-through and Fourier columns, the reference's hash features have
            # none (hashGridEmbedding.py:86) - is a second autograd node with hand-written first / second order
            # kernels.  The table node stays OFF the x-graph: autograd.grad(e, x, create_graph=True) would otherwise
            # run its backward (a dense scatter that is then thrown away - a custom Function cannot see that only
            # d/dx was asked for), and a sparse gradient collector would record a bogus contribution
            out = ops.embed_row_input_grad(x2, self.table, B, self.desc, self.grad_collector)
        else:
            four = self.freq_encoding(x2.float())
            xh = x2.detach() if self.frac_mode == "reference" else x2
            feat = ops.hash_features(xh, self.table, self.desc, fm, self.grad_collector)
            out = torch.cat([four, feat], dim=-1)
        return out.reshape(*lead, self.embeddings_dim)
