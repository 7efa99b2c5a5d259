"""Custom_Embedding_Network - the embedder registry (plugin point #2) with the reference's
constructor signature, registry keys, kwargs mapping and error behaviour
(reference: code/model/custom_embedder_decoder.py:13-164).

Registry coverage: 'HashGrid', 'FFB', 'StyleModNFFB' (HIP hash-grid kernels + HIP GEMM),
'FourierFeatures', 'NerfPos' (elementwise torch expressions, exactly the reference's).  The
tiny-cuda-nn variants ('HashGridTcnn', 'FFBTcnn') are out of scope by the north star ("not
tiny-cuda-nn recompiled") and raise with a message that says so.
"""
import torch
import torch.nn as nn

from .embeddings.frequency_enc import FourierFeature, PositionalEncoding
from .embeddings.hashGridEmbedding import MultiResHashGridMLP
from .embeddings.nffb3d import FourierFilterBanks

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")

_NOT_BUILT = {
    'HashGridTcnn': "tiny-cuda-nn backed encoder (excluded: no tcnn on MI355X)",
    'FFBTcnn': "tiny-cuda-nn backed encoder (excluded: no tcnn on MI355X)",
}


class Custom_Embedding_Network(nn.Module):
    def __init__(self, input_dims, network_dims, embed_type, multires, log2_max_hash_size, max_points_per_entry,
                 base_resolution, desired_resolution, bound):
        super().__init__()
        embed_kwargs = {
            'hash_grid_encoder_pytorch': {
                'include_input': True,
                'in_dim': input_dims,
                'n_levels': multires,
                'max_points_per_level': max_points_per_entry,
                'log2_hashmap_size': log2_max_hash_size,
                'base_resolution': base_resolution,
                'desired_resolution': desired_resolution,
            },
            'FourierFeature': {
                'num_channels': network_dims[0],
                'sigma': 1.0,
                'input_dims': input_dims,
                'include_input': True,
            },
            'fourier_filter_banks': {
                'GridEncoderNetConfig': {
                    'include_input': True, 'in_dim': input_dims, 'embed_type': 'HashGridTcnn',
                    'network_dims': network_dims, 'n_levels': multires,
                    'max_points_per_level': max_points_per_entry, 'log2_hashmap_size': log2_max_hash_size,
                    'base_resolution': base_resolution, 'desired_resolution': desired_resolution,
                    "base_sigma": 10.0, "exp_sigma": 1.26, "grid_embedding_std": 0.001, 'per_level_scale': 2.0,
                },
                'freq_enc_type': 'PositionalEncodingNET',
                'has_out': False,
                'bound': bound,
                'layers_type': 'SIREN',
            },
            'StyleModulatedNFFB': {
                'GridEncoderNetConfig': {
                    'include_input': True, 'in_dim': input_dims, 'embed_type': 'HashGridTcnn',
                    'network_dims': network_dims, 'n_levels': multires,
                    'max_points_per_level': max_points_per_entry, 'log2_hashmap_size': log2_max_hash_size,
                    'base_resolution': base_resolution, 'desired_resolution': desired_resolution,
                    "base_sigma": 10.0, "exp_sigma": 1.26, "grid_embedding_std": 0.001, 'per_level_scale': 2.0,
                },
                'freq_enc_type': 'PositionalEncodingNET',
                'has_out': False,
                'bound': bound,
                'layers_type': 'SIREN',
                'style_modulation': True,
            },
            'positional_encoding': {
                'include_input': True,
                'input_dims': input_dims,
                'max_freq_log2': log2_max_hash_size,
                'num_freqs': multires,
                'log_sampling': True,
                'periodic_fns': [torch.sin, torch.cos],
            },
        }
        embed_models = {
            'HashGrid': (MultiResHashGridMLP, 'hash_grid_encoder_pytorch'),
            'FFB': (FourierFilterBanks, 'fourier_filter_banks'),
            'StyleModNFFB': (FourierFilterBanks, 'StyleModulatedNFFB'),
            'NerfPos': (PositionalEncoding, 'positional_encoding'),
            'FourierFeatures': (FourierFeature, 'FourierFeature'),
        }
        if embed_type in _NOT_BUILT:
            raise NotImplementedError(f"embed_type {embed_type!r}: {_NOT_BUILT[embed_type]}")
        if embed_type not in embed_models:
            raise ValueError("Not a valid embedding model type")
        EmbedderClass, model_key = embed_models[embed_type]
        self.embedder_obj = EmbedderClass(**embed_kwargs[model_key])
        self.embeddings_dim = self.embedder_obj.embeddings_dim

    def forward(self, x, compute_grad=False):
        return self.embedder_obj.forward(x)
