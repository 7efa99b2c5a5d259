"""ImplicitNetwork / RenderingNetwork / IDRNetwork (plugin point #1) with the reference's
constructor arguments, attribute names, forward contracts and state_dict keys
(reference: code/model/implicit_differentiable_renderer.py:11-329).

Where the arithmetic runs:
  * no-grad SDF evaluation (ray tracing, eval)  -> ONE fused HIP kernel: encode + 9 MFMA layers +
    clamp (csrc/hm_sdf.hip via ops.sdf_fwd);
  * grad-enabled evaluation (training forward, gradient() with create_graph=True, rendering MLP)
    -> hash encoder kernels (csrc/hm_encode.hip) + the exact-fp32 MFMA GEMM (csrc/hm_gemm.hip)
    wrapped as an any-order differentiable op (ops.linear); activations/concats are elementwise
    torch expressions so autograd can differentiate twice.
nn.Linear + nn.utils.weight_norm are used purely as parameter containers (same keys
``lin{l}.weight_g / weight_v / bias`` and the same initialisation RNG stream as the reference).
"""
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, mlp_grad, ops
from ..utils import rend_util
from .custom_embedder_decoder import Custom_Embedding_Network
from .density_net import LaplaceDensity
from .embeddings.frequency_enc import get_embedder
from .embeddings.hashGridEmbedding import MultiResHashGridMLP
from .ray_tracing import RayTracing
from .sample_network import SampleNetwork


def _weight_normed(lin):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.utils.weight_norm(lin)


def _fold_all(nets, cache):
    """weight-norm fold of every Linear of `nets` in ONE launch (ops.weight_norm_fold; its backward is one launch too),
    left in `cache` for _folded_weight - torch._weight_norm is one launch per layer and pass (23 + 14 per step)"""
    lins = []
    for net in nets:
        for l in range(net.num_layers - 1):
            lin = getattr(net, "lin" + str(l))
            if hasattr(lin, "weight_g") and lin.weight_v.is_cuda and lin.weight_v.dtype == torch.float32:
                lins.append(lin)
    if lins and torch.is_grad_enabled():
        for lin, w in zip(lins, ops.weight_norm_fold([lin.weight_v for lin in lins], [lin.weight_g for lin in lins])):
            cache[id(lin)] = w


def _folded_weight(lin, cache=None):
    """W = g * v / ||v||_row for weight-normed layers (nn.utils.weight_norm, dim=0), else lin.weight.
    `cache` (a dict living for ONE IDRNetwork.forward) shares the fold between the evaluations of a step."""
    if not hasattr(lin, "weight_g"):
        return lin.weight
    if cache is not None and torch.is_grad_enabled():
        w = cache.get(id(lin))
        if w is None:
            w = cache[id(lin)] = torch._weight_norm(lin.weight_v, lin.weight_g, 0)
        return w
    return torch._weight_norm(lin.weight_v, lin.weight_g, 0)


class ImplicitNetwork(nn.Module):
    def __init__(self, feature_vector_size, d_in, d_out, dims, geometric_init=True, bias=1.0, skip_in=(),
                 weight_norm=True, multires=0, embed_type=None, log2_max_hash_size=10, max_points_per_entry=2,
                 base_resolution=64, desired_resolution=None, bound: float = 1.0):
        super().__init__()
        dims = [d_in] + list(dims) + [d_out + feature_vector_size]
        self.embed_fn = None
        self.embed_type = embed_type
        self.multires = multires
        self.dencity_net = LaplaceDensity(params_init={'beta': 0.9})  # (sic) attribute name is API
        if embed_type:
            if multires > 0:
                self.embed_model = Custom_Embedding_Network(
                    input_dims=d_in, network_dims=dims, embed_type=embed_type, multires=multires,
                    log2_max_hash_size=log2_max_hash_size, max_points_per_entry=max_points_per_entry,
                    base_resolution=base_resolution, desired_resolution=desired_resolution, bound=bound)
                self.embed_fn = self.embed_model.forward
                dims[0] = self.embed_model.embeddings_dim
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.dims = dims
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                # IDR / IGR geometric initialisation: the freshly built MLP approximates the SDF of a sphere
                if l == self.num_layers - 2:
                    torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    torch.nn.init.constant_(lin.bias, -bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            if weight_norm:
                lin = _weight_normed(lin)
            setattr(self, "lin" + str(l), lin)
        self.softplus = nn.Softplus(beta=100)
        for p in self.parameters():
            p.requires_grad = True
        self.use_fused_mlp_grad = True
        self._packed = None
        self._packed_key = None
        self._force_repack = False
        self._fold_cache = None
        self.sdf_tile_points = 0  # fused kernel tile: 0 auto (16-point tiles for small batches), 16, 64
        # BASELINE configs[4] ("bf16"): the ray tracer's coarse scans (100-sample sign-change search, closest approach)
        # run on the bf16 MFMA variant of the fused kernel; sphere tracing, secant refinement and every grad-enabled
        # evaluation stay exact fp32.  Off by default: the reference has no reduced-precision behaviour.
        self.bf16_coarse_search = False
        # precision of the same coarse scans on the SPLIT-operand kernel (csrc/hm_sdf_split.hip): None (off), "bf16x2"
        # (hi + lo bf16 operands, 16 significant bits: the accurate form of the "bf16" configuration) or "f16x2" (hi +
        # lo fp16, 22 significant bits: error below the rounding noise of an fp32 accumulation).  Takes precedence over
        # bf16_coarse_search.  Off by default, like it.
        self.coarse_split = None

    def coarse_mode(self):
        """hm_trace_cfg.coarse_bf16 value of this network: 0 exact fp32, 1 bf16, 2 split operands"""
        if self.coarse_split is not None:
            return 2
        return 1 if self.bf16_coarse_search else 0

    def __getstate__(self):  # the packed-weight cache holds raw device pointers: never copied / pickled
        d = self.__dict__.copy()
        d["_packed"], d["_packed_key"], d["_fold_cache"] = None, None, None
        return d

    # ---- fused no-grad path ---------------------------------------------------------------
    def _hash_embedder(self):
        emb = getattr(self, "embed_model", None)
        emb = getattr(emb, "embedder_obj", None)
        return emb if isinstance(emb, MultiResHashGridMLP) else None

    def _nffb_embedder(self):
        """the FourierFilterBanks module ('FFB' / 'StyleModNFFB') when its fused kernel covers the configuration"""
        emb = getattr(getattr(self, "embed_model", None), "embedder_obj", None)
        if emb is not None and type(emb).__name__ == "FourierFilterBanks" and emb._fused_ok():
            return emb
        return None

    def _lin_params(self):
        ps = []
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin" + str(l))
            ps += [lin.weight_v, lin.weight_g, lin.bias] if hasattr(lin, "weight_g") else [lin.weight, lin.bias]
        return ps

    def _fusable(self):
        emb = self._hash_embedder()
        if emb is None:
            nf = self._nffb_embedder()
            emb = nf.grid_enc if nf is not None else None
        if emb is None or not emb.table.is_cuda or emb.n_features != 2:
            return False
        widths = [getattr(self, "lin" + str(l)).bias.shape[0] for l in range(self.num_layers - 1)]
        return max(widths) <= 512 and 0 not in self.skip_in and (self.num_layers - 1) <= 16

    def packed_weights(self):
        """Fold weight-norm and pack the MFMA operand images; cached until a parameter changes."""
        ps = self._lin_params() + [self.dencity_net.beta]
        # _version does not see writes through raw pointers (training.optim.ClipAdam, graph replays): those bump
        # _lib.param_epoch(), which is part of the key
        key = (_lib.param_epoch(),) + tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or key != self._packed_key or self._force_repack or \
                self._packed.has_bf16 != bool(self.bf16_coarse_search) or self._packed.split != self.coarse_split:
            self._force_repack = False
            with torch.no_grad():
                fc = self._fold_cache or {}      # this forward's folds (made with grad enabled): reuse their values
                Ws = []
                for l in range(self.num_layers - 1):
                    lin = getattr(self, "lin" + str(l))
                    w = fc.get(id(lin))
                    Ws.append(w.detach() if w is not None else _folded_weight(lin))
                bs = [getattr(self, "lin" + str(l)).bias for l in range(self.num_layers - 1)]
                if self._packed is None or self._packed.bufs[0][0].device != Ws[0].device or \
                        self._packed.has_bf16 != bool(self.bf16_coarse_search) or self._packed.split != self.coarse_split:
                    self._packed = ops.PackedSdf(Ws, bs, self.dims[0], self.skip_in, self._beta_value(),
                                                 with_bf16=bool(self.bf16_coarse_search), split=self.coarse_split)
                else:
                    self._packed.update(Ws, bs, self._beta_value())
            self._packed_key = key
        return self._packed

    def _beta_value(self):
        """|beta| + beta_min as a Python float; read back from the device only when the parameter changed
        (it never receives a gradient), so re-packing inside a captured graph does not synchronise."""
        b = self.dencity_net.beta
        key = (b.data_ptr(), b._version)
        if getattr(self, "_beta_cache", (None, None))[0] != key:
            self._beta_cache = (key, float(self.dencity_net.get_beta().detach()))
        return self._beta_cache[1]

    def _fused(self, x, sdf_only, tile_points=None):
        tile = self.sdf_tile_points if tile_points is None else tile_points
        emb = self._hash_embedder()
        if emb is None:     # filter-bank embedder: its own fused kernel, then the MLP kernel on the embedding rows
            e = ops.nffb_fwd(self._nffb_embedder(), x)
            return ops.sdf_fwd_emb(self.packed_weights(), e, sdf_only=sdf_only, tile_points=tile)
        return ops.sdf_fwd(emb.desc, self.packed_weights(), x, emb.table.detach(), emb.freq_encoding.B,
                           ops.FRAC_MODES[emb.frac_mode], sdf_only=sdf_only, tile_points=tile)

    def sdf(self, x, tile_points=None):
        """no-grad SDF values [N] - the callable handed to RayTracing (reference passes
        ``lambda x: self.implicit_network(x)[:, 0]``, implicit_differentiable_renderer.py:257).
        tile_points: tile size of the fused kernel for this call (default: self.sdf_tile_points)."""
        with torch.no_grad():
            if self._fusable():
                return self._fused(x, True, tile_points)
            return self.forward(x)[:, 0]

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, input, compute_grad=False):
        needs_graph = torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not needs_graph and self._fusable():
            return self._fused(input, False)

        emb = self.embed_fn(input) if self.embed_fn is not None else input
        x = emb
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin" + str(l))
            if l in self.skip_in:
                x = torch.cat([x, emb], 1) / np.sqrt(2)
            x = ops.linear(x, _folded_weight(lin, self._fold_cache), lin.bias)
            if l < self.num_layers - 2:
                x = ops.softplus(x, self.softplus.beta, self.softplus.threshold)
        # soft clamp of the SDF column: tanh(s / (2 + LaplaceDensity(s))), density under no_grad
        s = x[..., 0]
        s = torch.tanh(s / (2 + self.dencity_net(s)))
        return torch.cat([s.unsqueeze(-1), x[..., 1:]], dim=-1)

    def gradient(self, x):
        """d sdf / d x with the graph kept (create_graph=True) -> [N,1,3]."""
        return self.forward_with_gradient(x)[1]

    def forward_with_gradient(self, x, cache_out=None, reuse=None):
        """(forward(x) [N,1+fvs], d sdf/d x [N,1,3]) from ONE network evaluation; the reference
        evaluates the network twice for this pair (get_rbg_value, :321-323) with identical values.
        cache_out (dict): receives the MLP node's saved tensors; reuse = (cache, row0, n): x holds the same numbers as
        rows [row0, row0 + n) of the batch that filled `cache` in this forward pass - the MLP forward is not recomputed
        (mlp_grad.sdf_mlp_rows), gradients flow through this call's own embedding as usual.

        Default route: the MLP and its input-gradient are one autograd node with an analytic backward
        (mlp_grad.sdf_mlp); only the embedding's Jacobian goes through autograd.  `use_fused_mlp_grad =
        False` selects the generic route (autograd.grad with create_graph over the layer ops)."""
        x.requires_grad_(True)
        if self.use_fused_mlp_grad and x.is_cuda and torch.is_grad_enabled() and len(self.skip_in) <= 1:
            e = self.embed_fn(x) if self.embed_fn is not None else x
            lins = [getattr(self, "lin" + str(l)) for l in range(self.num_layers - 1)]
            Ws = [_folded_weight(lin, self._fold_cache) for lin in lins]
            bs = [lin.bias for lin in lins]
            skip = self.skip_in[0] if self.skip_in else -1
            if reuse is not None and reuse[0]:
                out, g_e = mlp_grad.sdf_mlp_rows(e, reuse[0], reuse[1], reuse[2], Ws, bs)
            else:
                out, g_e = mlp_grad.sdf_mlp(e, Ws, bs, skip, self.softplus.beta, self.softplus.threshold,
                                            self._beta_value(), cache_out=cache_out)
            if e is x:
                g = g_e
            else:
                with ops.input_grad_only():
                    (g,) = torch.autograd.grad(e, x, g_e, create_graph=True, retain_graph=True)
            return out, g.unsqueeze(1)
        out = self.forward(x)
        y = out[:, :1]
        d_output = torch.ones_like(y, requires_grad=False, device=y.device)
        gradients = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=d_output, create_graph=True,
                                        retain_graph=True, only_inputs=True)[0]
        return out, gradients.unsqueeze(1)


class RenderingNetwork(nn.Module):
    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0,
                 viewdirs_embed_type='NerfPos'):
        super().__init__()
        self.feature_vector_size = feature_vector_size
        self.mode = mode
        dims = [d_in + feature_vector_size] + list(dims) + [d_out]
        self.multires_view = multires_view
        self.d_in = d_in
        self.embedview_fn = None
        if viewdirs_embed_type == 'SHEncoder':
            raise NotImplementedError("viewdirs_embed_type 'SHEncoder' is outside the hot-path scope (SURVEY.md 2, row 2)")
        elif viewdirs_embed_type == 'NerfPos':
            if multires_view > 0 and self.mode == 'idr':
                self.embedview_fn, input_ch = get_embedder(multires_view)
                dims[0] += input_ch
        elif viewdirs_embed_type in ('HashGrid', 'FFB', 'StyleModNFFB', 'FourierFeatures', 'HashGridCUDA', 'FFBTcnn',
                                     'HashGridTcnn'):
            if multires_view > 0 and self.mode == 'idr':
                d_in = 3
                self.embed_model = Custom_Embedding_Network(
                    input_dims=d_in, network_dims=dims, embed_type=viewdirs_embed_type, multires=multires_view,
                    max_points_per_entry=2, log2_max_hash_size=multires_view - 1, base_resolution=16,
                    desired_resolution=512, bound=1.0)
                self.embedview_fn = self.embed_model.forward
                dims[0] += (self.embed_model.embeddings_dim - d_in)
        else:
            raise ValueError('No Embedding Network config provided for VIEWDIRS')
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                lin = _weight_normed(lin)
            setattr(self, "lin" + str(l), lin)
        self.relu = nn.ReLU()
        self.tanh = nn.Tanh()
        self.use_fused_mlp = True    # Linear / ReLU stack as one autograd node (first-order); False: generic ops
        self._fold_cache = None
        for p in self.parameters():
            p.requires_grad = True

    def forward(self, points, normals, view_dirs, feature_vectors):
        if self.embedview_fn is not None:
            view_dirs = self.embedview_fn(view_dirs)
        if self.mode == 'idr':
            x = torch.cat([points, view_dirs, normals, feature_vectors], dim=-1)
        elif self.mode == 'no_view_dir':
            x = torch.cat([points, normals, feature_vectors], dim=-1)
        elif self.mode == 'no_normal':
            x = torch.cat([points, view_dirs, feature_vectors], dim=-1)
        lins = [getattr(self, "lin" + str(l)) for l in range(self.num_layers - 1)]
        if self.use_fused_mlp and x.is_cuda and torch.is_grad_enabled():
            # one first-order autograd node for the whole Linear / ReLU stack (mlp_grad.relu_mlp)
            x = mlp_grad.relu_mlp(x, [_folded_weight(lin, self._fold_cache) for lin in lins], [lin.bias for lin in lins])
            return self.tanh(x)
        for l, lin in enumerate(lins):
            x = ops.linear(x, _folded_weight(lin, self._fold_cache), lin.bias)
            if l < self.num_layers - 2:
                x = self.relu(x)
        return self.tanh(x)


class IDRNetwork(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.feature_vector_size = conf.get_int('feature_vector_size')
        implicit_kwargs = dict(conf.get_config('implicit_network'))
        if conf.get_config('embedding_network') is not None:
            implicit_kwargs.update(dict(conf.get_config('embedding_network')))
        self.implicit_network = ImplicitNetwork(self.feature_vector_size, **implicit_kwargs)
        self.rendering_network = RenderingNetwork(self.feature_vector_size, **conf.get_config('rendering_network'))
        self.ray_tracer = RayTracing(**conf.get_config('ray_tracer'))
        self.sample_network = SampleNetwork()
        self.object_bounding_sphere = conf.get_float('ray_tracer.object_bounding_sphere')
        # one SDF-network evaluation per training forward instead of the reference's three (identical
        # values; only legal while the ray points carry no camera gradient - checked per call)
        self.merge_evaluations = True
        # forward_static: the second SDF evaluation of the ray points reuses the first one's MLP forward (same inputs in
        # value); False re-evaluates them (A/B, tests)
        self.reuse_ray_rows = True
        # forward_static: camera rays + sphere intersections from one kernel (ops.camera_rays); False / HM_FUSED_CAMERA=0:
        # the torch expressions of utils/rend_util.py
        import os
        self.fused_camera = os.environ.get("HM_FUSED_CAMERA", "1") != "0"

    def forward(self, input):
        cache = {}
        self.implicit_network._fold_cache = cache
        self.rendering_network._fold_cache = cache
        try:
            if self.training:
                _fold_all((self.implicit_network, self.rendering_network), cache)
            return self._forward(input)
        finally:
            self.implicit_network._fold_cache = None
            self.rendering_network._fold_cache = None

    def _forward(self, input):
        intrinsics = input["intrinsics"]
        uv = input["uv"]
        pose = input["pose"]
        object_mask = input["object_mask"].reshape(-1)
        dev = uv.device

        ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
        batch_size, num_pixels, _ = ray_dirs.shape

        # 1. where does every ray meet the current surface? (no grad; the fused-kernel hot loop)
        self.implicit_network.eval()
        with torch.no_grad():
            points, network_object_mask, dists = self.ray_tracer(sdf=self.implicit_network.sdf, cam_loc=cam_loc,
                                                                 object_mask=object_mask, ray_directions=ray_dirs)
        self.implicit_network.train()

        # 2. re-express the hit points through (possibly learnable) camera parameters
        points = (cam_loc.unsqueeze(1) + dists.reshape(batch_size, num_pixels, 1) * ray_dirs).reshape(-1, 3)
        ray_dirs = ray_dirs.reshape(-1, 3)
        merged = self.training and self.merge_evaluations and torch.is_grad_enabled() and not points.requires_grad

        if merged:
            # Fixed cameras: the reference evaluates the SDF network on the ray points three times
            # (:264 all rays, :286 surface subset, :289 again inside gradient()) plus the eikonal samples.
            # The rows are independent, so ONE evaluation of [ray points ; eikonal samples] with
            # create_graph yields the same sdf_output, surface values, surface gradients and grad_theta.
            n_rays = batch_size * num_pixels
            bb = self.object_bounding_sphere
            eikonal_points = torch.empty(n_rays // 2, 3).uniform_(-bb, bb).to(dev)  # global CPU RNG, as the reference
            x_all = torch.cat([points.detach(), eikonal_points], 0)
            out_all, g_all = self.implicit_network.forward_with_gradient(x_all)
            sdf_output = out_all[:n_rays, 0:1]
            surface_mask = network_object_mask & object_mask
            surface_output = sdf_output[surface_mask]
            surface_sdf_values = surface_output.detach()
            surface_points_grad = g_all[:n_rays, 0, :][surface_mask].clone().detach()
            grad_theta = torch.cat([g_all[n_rays:, 0, :], g_all[:n_rays, 0, :]], 0)  # reference row order (:284,:291)
            differentiable_surface_points = self.sample_network(
                surface_output, surface_sdf_values, surface_points_grad, dists[surface_mask].unsqueeze(-1),
                cam_loc.unsqueeze(1).repeat(1, num_pixels, 1).reshape(-1, 3)[surface_mask], ray_dirs[surface_mask])
        elif self.training:
            sdf_output = self.implicit_network(points)[:, 0:1]
            surface_mask = network_object_mask & object_mask
            surface_points = points[surface_mask]
            surface_dists = dists[surface_mask].unsqueeze(-1)
            surface_ray_dirs = ray_dirs[surface_mask]
            surface_cam_loc = cam_loc.unsqueeze(1).repeat(1, num_pixels, 1).reshape(-1, 3)[surface_mask]
            surface_output = sdf_output[surface_mask]
            N = surface_points.shape[0]

            # eikonal samples: N/2 uniform in the bounding box (global CPU RNG, like the reference) + the ray points
            bb = self.object_bounding_sphere
            n_eik = batch_size * num_pixels // 2
            eikonal_points = torch.empty(n_eik, 3).uniform_(-bb, bb).to(dev)
            eikonal_points = torch.cat([eikonal_points, points.clone().detach()], 0)
            points_all = torch.cat([surface_points, eikonal_points], dim=0)

            output = self.implicit_network(surface_points)
            surface_sdf_values = output[:N, 0:1].detach()

            g = self.implicit_network.gradient(points_all)
            surface_points_grad = g[:N, 0, :].clone().detach()
            grad_theta = g[N:, 0, :]
            differentiable_surface_points = self.sample_network(surface_output, surface_sdf_values,
                                                                surface_points_grad, surface_dists,
                                                                surface_cam_loc, surface_ray_dirs)
        else:
            sdf_output = self.implicit_network(points)[:, 0:1]
            surface_mask = network_object_mask
            differentiable_surface_points = points[surface_mask]
            grad_theta = None

        view = -ray_dirs[surface_mask]
        rgb_values = torch.ones_like(points).float()
        if differentiable_surface_points.shape[0] > 0:
            rgb_values[surface_mask] = self.get_rbg_value(differentiable_surface_points, view)

        return {
            'points': points,
            'rgb_values': rgb_values,
            'sdf_output': sdf_output,
            'network_object_mask': network_object_mask,
            'object_mask': object_mask,
            'grad_theta': grad_theta,
        }

    def forward_static(self, input, eikonal_points, steps_u=None):
        """Training forward with STATIC shapes and no host synchronisation (fixed cameras only):
        the same quantities as forward(), but every per-ray selection is a mask instead of a
        boolean gather, so the step can be captured in a HIP graph (training/graph_step.py).

        Differences in mechanism: the rendering branch is evaluated for all rays and masked (the
        reference gathers the surface rays first); SampleNetwork's denominator is replaced by 1 on
        non-surface rays (their numerator is exactly 0, so the point is unchanged and finite).
        eikonal_points [n_rays//2, 3] and steps_u [n_steps] are the two random draws the reference
        makes inside forward (:277, ray_tracing.py:277), passed in as device tensors."""
        cache = {}
        self.implicit_network._fold_cache = cache
        self.rendering_network._fold_cache = cache
        try:
            _fold_all((self.implicit_network, self.rendering_network), cache)
            uv, pose, intrinsics = input["uv"], input["pose"], input["intrinsics"]
            object_mask = input["object_mask"].reshape(-1)
            if pose.requires_grad:
                raise RuntimeError("forward_static needs fixed cameras (pose without gradient)")
            sphere = None
            if self.fused_camera and uv.is_cuda and pose.dim() == 3 and tuple(pose.shape[1:]) == (4, 4):
                # rays and their bounding-sphere intersections in ONE launch (ops.camera_rays) instead of the ~38
                # elementwise / bmm launches of rend_util.get_camera_params + get_sphere_intersection
                ray_dirs, cam_loc, t_sph, hit = ops.camera_rays(uv, pose, intrinsics, self.ray_tracer.object_bounding_sphere)
                sphere = (t_sph, hit)
            else:
                ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
            if ray_dirs.requires_grad or cam_loc.requires_grad:
                raise RuntimeError("forward_static needs fixed cameras (pose without gradient)")
            batch_size, num_pixels, _ = ray_dirs.shape
            n_rays = batch_size * num_pixels
            self.implicit_network.eval()
            old = (self.ray_tracer.steps_override, self.ray_tracer.sphere_override)
            self.ray_tracer.steps_override, self.ray_tracer.sphere_override = steps_u, sphere
            try:
                with torch.no_grad():
                    _, network_object_mask, dists = self.ray_tracer(
                        sdf=self.implicit_network.sdf, cam_loc=cam_loc, object_mask=object_mask,
                        ray_directions=ray_dirs)
            finally:
                self.ray_tracer.steps_override, self.ray_tracer.sphere_override = old
            self.implicit_network.train()
            points = (cam_loc.unsqueeze(1) + dists.reshape(batch_size, num_pixels, 1) * ray_dirs).reshape(-1, 3)
            ray_dirs = ray_dirs.reshape(-1, 3)
            cams = cam_loc.unsqueeze(1).repeat(1, num_pixels, 1).reshape(-1, 3)

            # the rows are independent, so the merged evaluation runs in the ORDER grad_theta wants
            # ([eikonal samples ; ray points], reference :284,:291): no re-ordering copy, and no row / select slicing
            # of graph tensors (their autograd backward is zeros + a contiguous copy_ = a MEMCPY graph node)
            n_eik = eikonal_points.shape[0]
            x_all = torch.cat([eikonal_points, points.detach()], 0)
            mlp_cache = {} if self.reuse_ray_rows else None
            out_all, g_all = self.implicit_network.forward_with_gradient(x_all, cache_out=mlp_cache)
            grad_theta = g_all.reshape(-1, 3)
            sdf_output = ops.take_block(out_all, n_eik, n_rays, 0, 1)
            surface_mask = network_object_mask & object_mask
            m = surface_mask.unsqueeze(-1)

            # SampleNetwork (sample_network.py:10-20) on every ray; denominators of non-surface rays -> 1
            grad0 = grad_theta.detach()[n_eik:]
            dot = (grad0 * ray_dirs).sum(-1, keepdim=True)
            dot = torch.where(m, dot, torch.ones_like(dot))
            t_theta = dists.unsqueeze(-1) - (sdf_output - sdf_output.detach()) / dot
            diff_points = cams + t_theta * ray_dirs

            # diff_points holds the same numbers as `points` (t_theta = dists - 0 / dot): the second evaluation takes the
            # first one's MLP results for those rows and keeps only its own gradient path (mlp_grad._SdfMlpRows)
            out2, g2 = self.implicit_network.forward_with_gradient(diff_points, reuse=(mlp_cache, n_eik, n_rays))
            rgb = self.rendering_network(diff_points, g2.reshape(-1, 3), -ray_dirs, out2[:, 1:])
            rgb_values = torch.where(m, rgb, torch.ones_like(rgb))
            return {
                'points': points, 'rgb_values': rgb_values, 'sdf_output': sdf_output,
                'network_object_mask': network_object_mask, 'object_mask': object_mask, 'grad_theta': grad_theta,
            }
        finally:
            self.implicit_network._fold_cache = None
            self.rendering_network._fold_cache = None

    def get_rbg_value(self, points, view_dirs):
        if torch.is_grad_enabled():
            output, g = self.implicit_network.forward_with_gradient(points)  # one evaluation instead of two
        else:
            output = self.implicit_network(points)
            with torch.enable_grad():
                g = self.implicit_network.gradient(points)
        normals = g[:, 0, :]
        feature_vectors = output[:, 1:]
        return self.rendering_network(points, normals, view_dirs, feature_vectors)
