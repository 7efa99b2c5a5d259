"""Seeded synthetic parameters / inputs shared by the golden generator and the tests.

Everything here is plain numpy on the legacy ``RandomState`` generator (bit-stable across
numpy versions), so a fixture only has to store *outputs*: the tests regenerate the very
same weights, tables and points from the seed that ``make_goldens.py`` fed to the reference.

Nothing in this file comes from the reference; the statistics merely imitate its
geometric initialisation (reference: code/model/implicit_differentiable_renderer.py:55-85)
so that the SDF has a zero level set inside the unit sphere.
"""
import math

import numpy as np

F32 = np.float32


def level_table(n_levels, log2_hashmap_size, base_resolution, desired_resolution, in_dim=3):
    """Host double-precision level table, same arithmetic as the reference constructor
    (code/model/embeddings/hashGridEmbedding.py:126-132)."""
    beta = math.exp((math.log(desired_resolution) - math.log(base_resolution)) / (n_levels - 1))
    res, rows = [], []
    for l in range(n_levels):
        r = math.floor(base_resolution * (beta ** l))
        res.append(r)
        rows.append(min(r ** in_dim, 2 ** log2_hashmap_size))
    return res, rows


def fourier_sigma(base_resolution, desired_resolution):
    # reference: hashGridEmbedding.py:141 (note: divides by base_resolution - 1)
    return (math.log(desired_resolution) - math.log(base_resolution)) / (base_resolution - 1)


CONFIGS = {
    # name: (L, T, base, desired)
    "C1": (8, 14, 16, 512),
    "C2": (16, 19, 16, 512),
    "C4": (16, 22, 16, 512),
    "shipped": (6, 5, 8, 512),
    "viewdir": (4, 3, 16, 512),
    "tiny": (4, 8, 4, 32),
}


def make_table(seed, total_rows, F=2, scale=1e-4):
    return np.random.RandomState(seed).uniform(-scale, scale, (total_rows, F)).astype(F32)


def make_fourier_B(seed, L, sigma):
    return (np.random.RandomState(seed).standard_normal((3, L)) * sigma).astype(F32)


def make_points(seed, n, lo=-1.0, hi=1.0):
    return np.random.RandomState(seed).uniform(lo, hi, (n, 3)).astype(F32)


def adversarial_points(res_list):
    """Points on/near voxel boundaries, negatives, |x|>1, zeros."""
    pts = [
        (0.0, 0.0, 0.0), (-0.0, 0.0, -0.0), (1.0, 1.0, 1.0), (-1.0, -1.0, -1.0),
        (0.25, -0.5, 0.75), (-0.999, 0.001, 0.5), (0.999999, -0.999999, 0.5),
        (1.5, -1.5, 2.0), (-3.0, 2.5, -2.25), (1e-7, -1e-7, 1e-3), (0.5, 0.5, 0.5),
        (-0.5, -0.5, -0.5), (1.0, -1.0, 0.0), (0.33333334, -0.6666667, 0.1),
    ]
    for r in res_list:
        for k in (1, 3, r // 2, r - 1, r):
            v = k / r
            pts.append((v, -v, v))
            pts.append((np.nextafter(F32(v), F32(2)), np.nextafter(F32(-v), F32(-2)), np.nextafter(F32(v), F32(0))))
    return np.asarray(pts, dtype=F32)


def sdf_dims(E, hidden, d_out_total, skip_in):
    """Per-layer (in, out) of the SDF MLP (reference: implicit_differentiable_renderer.py:31,55-61)."""
    dims = [E] + list(hidden) + [d_out_total]
    shapes = []
    for l in range(len(dims) - 1):
        out = dims[l + 1] - dims[0] if (l + 1) in skip_in else dims[l + 1]
        shapes.append((dims[l], out))
    return shapes


def make_sdf_params(seed, E, hidden=(512,) * 8, d_out_total=257, skip_in=(4,), bias=0.6,
                    perturb=0.0, g_jitter=0.0):
    """Weight-normed SDF MLP parameters with geometric-init statistics.

    perturb > 0 fills the blocks geometric init leaves at zero (lin0[:,3:], skip[:, -(E-3):])
    with N(0, perturb*std) so that the hash features influence the output and table
    gradients are non-zero ('unit, grad' row).
    g_jitter > 0 makes weight_g differ from ||v|| so the weight-norm fold is exercised.
    Returns a dict with the reference's state_dict key names (lin{l}.weight_v ...).
    """
    rs = np.random.RandomState(seed)
    shapes = sdf_dims(E, hidden, d_out_total, skip_in)
    n = len(shapes)
    out = {}
    for l, (din, dout) in enumerate(shapes):
        std = math.sqrt(2.0) / math.sqrt(dout)
        if l == n - 1:
            v = rs.normal(math.sqrt(math.pi) / math.sqrt(din), 1e-4, (dout, din))
            # feature rows (1..): ordinary small weights so the feature vector is not constant
            v[1:] = rs.normal(0.0, std, (dout - 1, din))
            b = np.full((dout,), 0.0)
            b[0] = -bias
        elif l == 0:
            v = np.zeros((dout, din))
            v[:, :3] = rs.normal(0.0, std, (dout, 3))
            if perturb > 0:
                v[:, 3:] = rs.normal(0.0, perturb * std, (dout, din - 3))
            b = np.zeros((dout,))
        elif l in skip_in:
            v = rs.normal(0.0, std, (dout, din))
            if perturb > 0:
                v[:, -(E - 3):] = rs.normal(0.0, perturb * std, (dout, E - 3))
            else:
                v[:, -(E - 3):] = 0.0
            b = np.zeros((dout,))
        else:
            v = rs.normal(0.0, std, (dout, din))
            b = np.zeros((dout,))
        if perturb > 0:
            b = b + rs.normal(0.0, 0.01, b.shape)
        g = np.sqrt((v * v).sum(axis=1, keepdims=True))
        if g_jitter > 0:
            g = g * rs.uniform(1.0 - g_jitter, 1.0 + g_jitter, g.shape)
        out[f"lin{l}.weight_v"] = v.astype(F32)
        out[f"lin{l}.weight_g"] = g.astype(F32)
        out[f"lin{l}.bias"] = b.astype(F32)
    return out


def make_render_params(seed, d_in0=281, hidden=(512,) * 4, d_out=3, g_jitter=0.1):
    """Rendering MLP parameters (reference: implicit_differentiable_renderer.py:186-196)."""
    rs = np.random.RandomState(seed)
    dims = [d_in0] + list(hidden) + [d_out]
    out = {}
    for l in range(len(dims) - 1):
        din, dout = dims[l], dims[l + 1]
        bound = 1.0 / math.sqrt(din)
        v = rs.uniform(-bound, bound, (dout, din))
        b = rs.uniform(-bound, bound, (dout,))
        g = np.sqrt((v * v).sum(axis=1, keepdims=True)) * rs.uniform(1 - g_jitter, 1 + g_jitter, (dout, 1))
        out[f"lin{l}.weight_v"] = v.astype(F32)
        out[f"lin{l}.weight_g"] = g.astype(F32)
        out[f"lin{l}.bias"] = b.astype(F32)
    return out


def make_nffb_params(seed, L, style, table_scale=0.5, F=2, log2_T=5, base=16, desired=512):
    """Parameters of a FourierFilterBanks embedder ('FFB' / 'StyleModNFFB') with SIREN-like statistics, keyed as the
    reference's state_dict (embedder_obj.*): ff_lin{l}, out_layer, [StyleAttentionBlock.*], grid_enc levels + B."""
    rs = np.random.RandomState(seed)
    W = 8 + 8 * L
    w0 = float(L ** F - L)
    out = {}
    for l in range(L - 1):
        fan_in = 3 if l == 0 else W
        bound = 1.0 / fan_in if l == 0 else math.sqrt(6.0 / fan_in) / w0
        out[f"ff_lin{l}.weight"] = rs.uniform(-bound, bound, (W, fan_in)).astype(F32)
        out[f"ff_lin{l}.bias"] = rs.uniform(-bound, bound, (W,)).astype(F32)
    b = 1.0 / math.sqrt(W)
    out["out_layer.weight"] = rs.uniform(-b, b, (W, W)).astype(F32)
    out["out_layer.bias"] = rs.uniform(-b, b, (W,)).astype(F32)
    if style:
        out["StyleAttentionBlock.linear_transform.weight"] = rs.uniform(-b, b, (W, W)).astype(F32)
        out["StyleAttentionBlock.linear_transform.bias"] = rs.uniform(-b, b, (W,)).astype(F32)
        out["StyleAttentionBlock.attention.weight"] = rs.uniform(-0.5, 0.5, (1, 3)).astype(F32)
        out["StyleAttentionBlock.attention.bias"] = rs.uniform(-0.5, 0.5, (1,)).astype(F32)
    res, rows = level_table(L, log2_T, base, desired)
    for l in range(L):
        out[f"grid_enc.levels.{l}.embedding.weight"] = rs.uniform(-table_scale, table_scale, (rows[l], F)).astype(F32)
    out["grid_enc.freq_encoding.B"] = (rs.standard_normal((3, L)) * fourier_sigma(base, desired)).astype(F32)
    return out


def make_embedder_state(seed, cfg, table_scale=1e-4):
    """(per-level tables list, B) for one hash-grid embedder config."""
    L, T, b, d = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    res, rows = level_table(L, T, b, d)
    table = make_table(seed, int(sum(rows)), 2, table_scale)
    offs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    levels = [table[offs[l]:offs[l + 1]] for l in range(L)]
    B = make_fourier_B(seed + 1, L, fourier_sigma(b, d))
    return levels, B, res, rows


def make_rays(seed, n_rays, cam_radius=2.5, ball_radius=1.0):
    """'Uniform-sphere rays' of SURVEY.md section 8(d): one camera on the radius-2.5 sphere,
    targets uniform in the unit ball."""
    rs = np.random.RandomState(seed)
    g = rs.standard_normal(3)
    cam = (cam_radius * g / np.linalg.norm(g)).astype(F32)
    u = rs.standard_normal((n_rays, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    r = ball_radius * rs.uniform(0, 1, (n_rays, 1)) ** (1.0 / 3.0)
    p = u * r
    d = p - cam[None, :].astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return cam.reshape(1, 3), d.astype(F32).reshape(1, n_rays, 3)
