"""torch-CPU fp32 restatement of the grad-enabled hot path (TEST INFRASTRUCTURE ONLY).

Used (a) to pin floating-point gradients / the full training step on the CPU against the
fixtures generated from the reference, (b) as the stand-in model of the world_size-2 gloo test,
(c) as bench.py's cpu_baseline ("port": the same op sequence the reference's PyTorch path runs,
on the host cores).  Parity status: PINNED by tests/test_torch_ref_cpu.py against
tests/golden/sdf_*.npz and idr_step_C1.npz.

Arithmetic lives here (hash indices in int64 like the reference, F.linear GEMMs, softplus, the
clamp); the ray search, SampleNetwork and the camera helpers are oracle/ray_ref.py - nothing of the
product's control flow is borrowed (only IDRLoss, an elementwise formula pinned by the idr_step fixtures,
is shared by the tests).  Citations: /root/reference/code/model/...
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ray_ref

_PRIMES = torch.tensor([1, 3, 2654435761], dtype=torch.int64)  # embeddings/hashGridEmbedding.py:14
_CORNERS = torch.tensor([[(n >> d) & 1 for d in range(3)] for n in range(8)], dtype=torch.int64)


class RefHashGrid(nn.Module):
    """embeddings/hashGridEmbedding.py:81-102,150-155 + frequency_enc.py:63-67 in reference frac mode."""

    def __init__(self, res, rows, table, B):
        super().__init__()
        self.res, self.rows = [int(r) for r in res], [int(r) for r in rows]
        off = np.concatenate([[0], np.cumsum(self.rows)])
        self.tables = nn.ParameterList([nn.Parameter(table[off[l]:off[l + 1]].clone()) for l in range(len(res))])
        self.register_buffer("B", B.clone())

    def ids(self, x, l):
        xi = (x * self.res[l]).long()
        c = xi.unsqueeze(-2) + _CORNERS.to(x.device)            # [N,8,3]
        h = (c * _PRIMES.to(x.device)) & 0xffffffff
        return (h[..., 0] ^ h[..., 1] ^ h[..., 2]) % self.rows[l]

    def forward(self, x):
        xp = torch.matmul(2 * np.pi * x, self.B)
        parts = [x, torch.sin(xp), torch.cos(xp)]
        for l in range(len(self.res)):
            # weights are (1,0,...,0): xf = x - x.float() == 0 (hashGridEmbedding.py:86) -> corner 0 only
            parts.append(self.tables[l][self.ids(x, l)[:, 0]])
        return torch.cat(parts, -1)


def _fold(v, g):
    return torch._weight_norm(v, g, 0)


class RefImplicit(nn.Module):
    """implicit_differentiable_renderer.py:89-128"""

    def __init__(self, grid, lin_params, skip_in=(4,), beta=0.9):
        super().__init__()
        self.grid = grid
        self.n = len(lin_params)
        for l, (v, g, b) in enumerate(lin_params):
            self.register_parameter(f"v{l}", nn.Parameter(v.clone()))
            self.register_parameter(f"g{l}", nn.Parameter(g.clone()))
            self.register_parameter(f"b{l}", nn.Parameter(b.clone()))
        self.skip_in = tuple(skip_in)
        self.beta = float(abs(beta)) + 1e-4

    def forward(self, x):
        emb = self.grid(x)
        h = emb
        for l in range(self.n):
            if l in self.skip_in:
                h = torch.cat([h, emb], 1) / np.sqrt(2)
            h = F.linear(h, _fold(getattr(self, f"v{l}"), getattr(self, f"g{l}")), getattr(self, f"b{l}"))
            if l < self.n - 1:
                h = F.softplus(h, beta=100)
        s = h[..., 0]
        with torch.no_grad():
            beta = torch.tensor(0.9).abs() + torch.tensor(0.0001)
            rho = (1 / beta) * (0.5 + 0.5 * s.sign() * torch.expm1(-s.abs() / beta))
        s = torch.tanh(s / (2 + rho))
        return torch.cat([s.unsqueeze(-1), h[..., 1:]], -1)

    def sdf(self, x):
        with torch.no_grad():
            return self.forward(x)[:, 0]

    def gradient(self, x):
        x.requires_grad_(True)
        y = self.forward(x)[:, :1]
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True, retain_graph=True, only_inputs=True)[0]
        return g.unsqueeze(1)


class RefRendering(nn.Module):
    """implicit_differentiable_renderer.py:202-223 (mode 'idr', hash-grid view-dir embedder)"""

    def __init__(self, grid, lin_params):
        super().__init__()
        self.grid = grid
        self.n = len(lin_params)
        for l, (v, g, b) in enumerate(lin_params):
            self.register_parameter(f"v{l}", nn.Parameter(v.clone()))
            self.register_parameter(f"g{l}", nn.Parameter(g.clone()))
            self.register_parameter(f"b{l}", nn.Parameter(b.clone()))

    def forward(self, points, normals, view_dirs, feature_vectors):
        h = torch.cat([points, self.grid(view_dirs), normals, feature_vectors], -1)
        for l in range(self.n):
            h = F.linear(h, _fold(getattr(self, f"v{l}"), getattr(self, f"g{l}")), getattr(self, f"b{l}"))
            if l < self.n - 1:
                h = F.relu(h)
        return torch.tanh(h)


def _grid_from(emb):
    return RefHashGrid(emb.resolutions, emb.hashmap_sizes, emb.table.detach().cpu(), emb.freq_encoding.B.detach().cpu())


def _lins(net):
    out = []
    for l in range(net.num_layers - 1):
        lin = getattr(net, f"lin{l}")
        out.append((lin.weight_v.detach().cpu(), lin.weight_g.detach().cpu(), lin.bias.detach().cpu()))
    return out


class RefIDR(nn.Module):
    """implicit_differentiable_renderer.py:242-329 with CPU arithmetic; built from a package IDRNetwork's
    parameters (the package constructs on the CPU without touching the GPU)."""

    def __init__(self, model):
        super().__init__()
        imp, ren = model.implicit_network, model.rendering_network
        self.implicit_network = RefImplicit(_grid_from(imp.embed_model.embedder_obj), _lins(imp), imp.skip_in)
        self.rendering_network = RefRendering(_grid_from(ren.embed_model.embedder_obj), _lins(ren))
        rt = model.ray_tracer
        self.ray_tracer = ray_ref.RayTraceRef(rt.object_bounding_sphere, rt.sdf_threshold, rt.line_search_step,
                                              rt.line_step_iters, rt.sphere_tracing_iters, rt.n_steps,
                                              rt.n_secant_steps)
        self.object_bounding_sphere = model.object_bounding_sphere

    def forward(self, input):
        object_mask = input["object_mask"].reshape(-1)
        ray_dirs, cam_loc = ray_ref.camera_rays(input["uv"], input["pose"], input["intrinsics"])
        B, P, _ = ray_dirs.shape
        self.ray_tracer.training = self.training
        with torch.no_grad():
            points, net_mask, dists = self.ray_tracer(self.implicit_network.sdf, cam_loc, object_mask, ray_dirs)
        points = (cam_loc.unsqueeze(1) + dists.reshape(B, P, 1) * ray_dirs).reshape(-1, 3)
        sdf_output = self.implicit_network(points)[:, 0:1]
        ray_dirs = ray_dirs.reshape(-1, 3)
        if self.training:
            sm = net_mask & object_mask
            sp = points[sm]
            N = sp.shape[0]
            eik = torch.empty(B * P // 2, 3).uniform_(-self.object_bounding_sphere, self.object_bounding_sphere)
            pts_all = torch.cat([sp, eik, points.clone().detach()], 0)
            out_s = self.implicit_network(sp)
            g = self.implicit_network.gradient(pts_all)
            dsp = ray_ref.sample_point(sdf_output[sm], out_s[:N, 0:1].detach(), g[:N, 0, :].clone().detach(),
                                      dists[sm].unsqueeze(-1),
                                      cam_loc.unsqueeze(1).repeat(1, P, 1).reshape(-1, 3)[sm], ray_dirs[sm])
            grad_theta = g[N:, 0, :]
        else:
            sm = net_mask
            dsp = points[sm]
            grad_theta = None
        rgb = torch.ones_like(points).float()
        if dsp.shape[0] > 0:
            o = self.implicit_network(dsp)
            nrm = self.implicit_network.gradient(dsp)[:, 0, :]
            rgb[sm] = self.rendering_network(dsp, nrm, -ray_dirs[sm], o[:, 1:])
        return {'points': points, 'rgb_values': rgb, 'sdf_output': sdf_output, 'network_object_mask': net_mask,
                'object_mask': object_mask, 'grad_theta': grad_theta}
