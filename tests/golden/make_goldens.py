#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python
(imported read-only from /root/reference/code, CPU, fp32) on seeded synthetic inputs.

Runs ONLY in the build container (the reference never travels to the GPU box); the
resulting small .npz files are committed and are what pins oracle/ and the HIP path.
Recipe: SURVEY.md Appendix D.  Usage:  python tests/golden/make_goldens.py [name ...]

Fixtures hold data only (inputs, seeds, expected outputs) - no reference source text.
"""
import contextlib
import io
import os
import sys
import types

sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import params as P  # noqa: E402

for _name in ("tinycudann", "imageio", "skimage", "cv2"):  # absent here; IO helpers / tcnn wrappers only
    sys.modules[_name] = types.ModuleType(_name)
sys.path.insert(0, "/root/reference/code")
torch.Tensor.cuda = lambda self, *a, **k: self  # reference hard-codes .cuda(); no GPU here

with contextlib.redirect_stdout(io.StringIO()):
    from model.embeddings import hashGridEmbedding as ref_hg  # noqa: E402
    from model.implicit_differentiable_renderer import (IDRNetwork, ImplicitNetwork,  # noqa: E402
                                                        RenderingNetwork)
    from model.ray_tracing import RayTracing  # noqa: E402
    from model.loss import IDRLoss  # noqa: E402
    from utils import rend_util as ref_rend  # noqa: E402

torch.set_num_threads(8)
T = torch.from_numpy


class Conf(dict):
    """Stand-in for pyhocon's ConfigTree (pyhocon is not installed)."""

    def _g(self, k):
        d = self
        for p in k.split("."):
            d = d[p]
        return d

    def get_int(self, k):
        return int(self._g(k))

    def get_float(self, k):
        return float(self._g(k))

    def get_config(self, k):
        v = self.get(k)
        return Conf(v) if v is not None else None


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def ref_embedder(cfg):
    L, Tt, b, d = P.CONFIGS[cfg]
    return quiet(ref_hg.MultiResHashGridMLP, True, 3, L, 2, Tt, b, d)


def load_embedder(emb, levels, B):
    sd = {f"levels.{l}.embedding.weight": T(np.ascontiguousarray(t)) for l, t in enumerate(levels)}
    sd["freq_encoding.B"] = T(B)
    emb.load_state_dict(sd)


def ref_implicit(cfg, hidden, fvs, seed, perturb, table_scale, g_jitter=0.1, bias=0.6):
    L, Tt, b, d = P.CONFIGS[cfg]
    net = quiet(ImplicitNetwork, fvs, 3, 1, list(hidden), True, 0.6, [4], True, multires=L,
                embed_type="HashGrid", log2_max_hash_size=Tt, max_points_per_entry=2,
                base_resolution=b, desired_resolution=d, bound=1.0)
    levels, B, res, rows = P.make_embedder_state(seed, cfg, table_scale)
    load_embedder(net.embed_model.embedder_obj, levels, B)
    E = 3 + 2 * L + 2 * L
    prm = P.make_sdf_params(seed + 7, E, hidden, 1 + fvs, (4,), bias, perturb, g_jitter)
    sd = net.state_dict()
    for k, v in prm.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = T(v)
    net.load_state_dict(sd)
    return net


# ----------------------------------------------------------------------------------------
def gen_levels():
    out = {}
    for cfg in P.CONFIGS:
        emb = ref_embedder(cfg)
        out[cfg + "_res"] = np.asarray([int(l.resolution) for l in emb.levels], np.int64)
        out[cfg + "_rows"] = np.asarray([int(l.hashmap_size) for l in emb.levels], np.int64)
        out[cfg + "_E"] = np.asarray(emb.embeddings_dim, np.int64)
        out[cfg + "_B_shape"] = np.asarray(emb.freq_encoding.B.shape, np.int64)
    save("levels", **out)


def gen_hash_ids():
    combos = [(16, 4096), (20, 8000), (25, 15625), (101, 524288), (512, 524288),
              (512, 4194304), (8, 32), (322, 524288), (161, 4173281), (512, 8)]
    out = {"combos": np.asarray(combos, np.int64)}
    for i, (res, rows) in enumerate(combos):
        x = np.concatenate([P.make_points(100 + i, 1024, -1.25, 1.25), P.adversarial_points([res])])
        lvl = ref_hg._HashGridMLP(3, 2, rows, res)
        xt = T(x) * lvl.resolution
        xi = xt.long()
        inds = torch.where(lvl.bin_mask.reshape(1, 8, 3), xi.unsqueeze(-2), xi.unsqueeze(-2) + 1)
        ids = ref_hg.hash_func(inds, lvl.primes, lvl.hashmap_size)
        out[f"x_{i}"] = x
        out[f"xi_{i}"] = xi.numpy().astype(np.int32)
        out[f"ids_{i}"] = ids.numpy().astype(np.uint32)
    save("hash_ids", **out)


def gen_encode():
    for cfg, scale, seed in [("C1", 0.5, 11), ("C2", 1e-4, 12), ("shipped", 0.5, 13),
                             ("viewdir", 0.5, 14), ("tiny", 0.5, 15)]:
        emb = ref_embedder(cfg)
        levels, B, res, rows = P.make_embedder_state(seed, cfg, scale)
        load_embedder(emb, levels, B)
        x = np.concatenate([P.make_points(seed + 100, 1024 - 64, -1.1, 1.1), P.adversarial_points(res)[:64]])
        with torch.no_grad():
            y = emb(T(x)).numpy()
        save(f"encode_{cfg}", x=x, out=y, seed=np.int64(seed), table_scale=np.float64(scale))


def gen_encode_bwd():
    for cfg, seed in [("C1", 21), ("tiny", 22)]:
        emb = ref_embedder(cfg)
        levels, B, res, rows = P.make_embedder_state(seed, cfg, 0.5)
        load_embedder(emb, levels, B)
        x = P.make_points(seed + 100, 2048, -1.0, 1.0)
        # duplicates so several points scatter into one row
        x[1024:1536] = x[:512]
        y = emb(T(x))
        d_out = np.random.RandomState(seed + 5).standard_normal(y.shape).astype(np.float32)
        (y * T(d_out)).sum().backward()
        g = torch.cat([l.embedding.weight.grad for l in emb.levels], 0).numpy()
        nz = np.nonzero(np.abs(g).sum(1))[0]
        save(f"encode_bwd_{cfg}", x=x, d_out=d_out, nz_rows=nz.astype(np.int64), nz_grad=g[nz],
             total_rows=np.int64(g.shape[0]), seed=np.int64(seed))


def _sample_idx(shape, k, seed):
    n = int(np.prod(shape))
    return np.random.RandomState(seed).choice(n, size=min(k, n), replace=False)


def gen_sdf(only=None):
    """SDF MLP forward, gradient(), first-order and double-backward parameter grads."""
    # (a) full width, C1 embedder: outputs + gradient + grad norms / sampled entries
    for tag, cfg, hidden, fvs, perturb, scale, seed, n in [
        ("full", "C1", (512,) * 8, 256, 0.5, 0.5, 31, 128),
        ("init", "C1", (512,) * 8, 256, 0.0, 1e-4, 32, 128),
        ("narrow", "tiny", (64,) * 8, 16, 0.5, 0.5, 33, 256),
        # the benchmarked width: L=16 -> E=67, layer-3 width 445, skip K = 445 + 67
        ("C2", "C2", (512,) * 8, 256, 0.5, 0.5, 34, 256),
    ]:
        if only and tag not in only:
            continue
        net = ref_implicit(cfg, hidden, fvs, seed, perturb, scale)
        x = P.make_points(seed + 100, n, -1.0, 1.0)
        xt = T(x.copy())
        net.eval()
        with torch.no_grad():
            out = net(xt).numpy()
        net.train()
        # first-order: loss = sum(out * R)
        R = np.random.RandomState(seed + 3).standard_normal(out.shape).astype(np.float32)
        net.zero_grad()
        xg = T(x.copy()).requires_grad_(True)
        o = net(xg)
        (o * T(R)).sum().backward()
        first = {k: p.grad.clone().numpy() for k, p in net.named_parameters() if p.grad is not None}
        dx_first = xg.grad.numpy()
        # second-order: eikonal loss on gradient()
        net.zero_grad()
        xg2 = T(x.copy())
        g = net.gradient(xg2)  # [N,1,3], create_graph=True
        grad_np = g.detach().numpy()[:, 0, :]
        eik = ((g[:, 0, :].norm(2, dim=1) - 1) ** 2).mean()
        eik.backward()
        second = {k: p.grad.clone().numpy() for k, p in net.named_parameters() if p.grad is not None}
        arrays = dict(x=x, out=out, R=R, gradient=grad_np, dx_first=dx_first, eik=np.float64(eik.item()),
                      seed=np.int64(seed), perturb=np.float64(perturb), table_scale=np.float64(scale),
                      fvs=np.int64(fvs), hidden=np.asarray(hidden, np.int64))
        for label, grads in (("g1", first), ("g2", second)):
            for k, v in grads.items():
                if k.startswith("embed_model"):
                    lvl = k.split(".")[3] if "levels" in k else "B"
                    key = f"{label}:table{lvl}"
                else:
                    key = f"{label}:{k}"
                arrays[key + ":norm"] = np.float64(np.linalg.norm(v.astype(np.float64)))
                if tag == "narrow" or v.size <= 4096:
                    arrays[key + ":full"] = v
                else:
                    idx = _sample_idx(v.shape, 256, 5)
                    arrays[key + ":idx"] = idx.astype(np.int64)
                    arrays[key + ":val"] = v.reshape(-1)[idx]
        save(f"sdf_{tag}", **arrays)


def _record_uniform():
    draws = []
    orig = torch.Tensor.uniform_

    def rec(self, *a, **k):
        r = orig(self, *a, **k)
        draws.append(r.clone().numpy())
        return r

    torch.Tensor.uniform_ = rec
    return draws, orig


def gen_raytrace(only=None):
    # (C2: last-layer bias -1.0 instead of -0.6, so that the perturbed L=16 network keeps a surface: r ~ 0.6 .. 0.8)
    for tag, cfg, n, perturb, scale, seed, bias in [("init", "C1", 256, 0.0, 1e-4, 41, 0.6),
                                                    ("bumpy", "C1", 256, 0.1, 0.05, 44, 0.6),
                                                    ("C2", "C2", 2048, 0.1, 0.05, 45, 1.0)]:
        if only and tag not in only:
            continue
        net = ref_implicit(cfg, (512,) * 8, 256, seed, perturb, scale, bias=bias)
        net.eval()
        cam, dirs = P.make_rays(seed + 50, n)
        rs = np.random.RandomState(seed + 60)
        # some rays that miss the bounding sphere entirely
        miss = rs.choice(n, 24, replace=False)   # (24 of n rays, whatever n is)
        d = dirs[0].copy()
        d[miss[:12]] = -d[miss[:12]]  # behind the camera (t clamps to 0)
        side = np.cross(d[miss[12:]].astype(np.float64), cam[0].astype(np.float64))
        d[miss[12:]] = (side / np.linalg.norm(side, axis=1, keepdims=True)).astype(np.float32)  # true misses
        dirs = d.reshape(1, n, 3)
        object_mask = rs.uniform(0, 1, n) < 0.8
        sdf = lambda p: net(p)[:, 0]  # noqa: E731
        arrays = dict(cam_loc=cam, ray_dirs=dirs, object_mask=object_mask, seed=np.int64(seed),
                      perturb=np.float64(perturb), table_scale=np.float64(scale), bias=np.float64(bias))
        for mode in ("train", "eval"):
            rt = RayTracing(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5,
                            line_step_iters=3, sphere_tracing_iters=10, n_steps=100, n_secant_steps=8)
            rt.train(mode == "train")
            torch.manual_seed(seed)
            draws, orig = _record_uniform()
            calls = []

            def counted(p):
                calls.append(p.shape[0])
                return sdf(p)
            try:
                with torch.no_grad():
                    pts, mask, dists = quiet(rt, sdf=counted, cam_loc=T(cam), object_mask=T(object_mask),
                                             ray_directions=T(dirs))
            finally:
                torch.Tensor.uniform_ = orig
            arrays[f"{mode}_points"] = pts.numpy()
            arrays[f"{mode}_mask"] = mask.numpy()
            arrays[f"{mode}_dists"] = dists.numpy()
            arrays[f"{mode}_sdf_evals"] = np.int64(sum(calls))
            if mode == "train":
                assert len(draws) == 1, len(draws)
                arrays["steps"] = draws[0]
        # sphere intersection helper
        si, mi = ref_rend.get_sphere_intersection(T(cam), T(dirs), r=1.0)
        arrays["sphere_intersections"] = si.numpy()
        arrays["mask_intersect"] = mi.numpy()
        save(f"raytrace_{tag}", **arrays)


def idr_conf(cfg, hidden=(512,) * 8, fvs=256, rdims=(512,) * 4):
    L, Tt, b, d = P.CONFIGS[cfg]
    return Conf(
        feature_vector_size=fvs,
        implicit_network=dict(d_in=3, d_out=1, dims=list(hidden), geometric_init=True, bias=0.6, skip_in=[4],
                              weight_norm=True, multires=L),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, viewdirs_embed_type="HashGrid", dims=list(rdims),
                               weight_norm=True, multires_view=4),
        ray_tracer=dict(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=3,
                        sphere_tracing_iters=10, n_steps=100, n_secant_steps=8),
        embedding_network=dict(embed_type="HashGrid", log2_max_hash_size=Tt, max_points_per_entry=2,
                               base_resolution=b, desired_resolution=d, bound=1.0),
    )


def gen_idr_step(cfg="C1", n=256, seed=51, n_steps=3, bias=0.6):
    """Full IDRNetwork.forward + IDRLoss + backward + Adam steps: config-1 shape (256 rays, 3 steps) and the
    benchmarked config-2 shape (L=16, T=2^19, 2048 rays, 1 step; gen_idr_step_C2)."""
    model = quiet(IDRNetwork, idr_conf(cfg))
    # --- load seeded parameters
    L = P.CONFIGS[cfg][0]
    E = 3 + 4 * L
    levels, B, _, _ = P.make_embedder_state(seed, cfg, 0.05)
    load_embedder(model.implicit_network.embed_model.embedder_obj, levels, B)
    sd = model.implicit_network.state_dict()
    for k, v in P.make_sdf_params(seed + 7, E, (512,) * 8, 257, (4,), bias, 0.1, 0.1).items():
        sd[k] = T(v)
    model.implicit_network.load_state_dict(sd)
    vlevels, vB, _, _ = P.make_embedder_state(seed + 20, "viewdir", 0.5)
    load_embedder(model.rendering_network.embed_model.embedder_obj, vlevels, vB)
    sd = model.rendering_network.state_dict()
    for k, v in P.make_render_params(seed + 9).items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = T(v)
    model.rendering_network.load_state_dict(sd)
    # --- inputs: identity-K pinhole whose uv reproduce 'uniform-sphere' ray dirs
    cam, dirs = P.make_rays(seed + 50, n)
    # camera frame: z axis looks at origin
    z = -cam[0] / np.linalg.norm(cam[0])
    up = np.array([0.0, 1.0, 0.0])
    xax = np.cross(up, z)
    xax /= np.linalg.norm(xax)
    yax = np.cross(z, xax)
    R = np.stack([xax, yax, z], 1)  # columns = camera axes in world
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = cam[0]
    dc = dirs[0].astype(np.float64) @ R  # dirs in camera frame
    uv = (dc[:, :2] / dc[:, 2:3]).astype(np.float32).reshape(1, n, 2)
    intr = np.eye(4, dtype=np.float32).reshape(1, 4, 4)
    rs = np.random.RandomState(seed + 60)
    object_mask = (rs.uniform(0, 1, n) < 0.85).reshape(1, n)
    rgb_gt = rs.uniform(-1, 1, (1, n, 3)).astype(np.float32)
    inp = dict(intrinsics=T(intr), uv=T(uv), pose=T(pose.reshape(1, 4, 4)), object_mask=T(object_mask))
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    opt = torch.optim.Adam(model.parameters(), lr=1.0e-4)
    arrays = dict(intrinsics=intr, uv=uv, pose=pose.reshape(1, 4, 4), object_mask=object_mask, rgb_gt=rgb_gt,
                  seed=np.int64(seed), bias=np.float64(bias))
    model.train()
    names = [k for k, _ in model.named_parameters()]
    for step in range(n_steps):
        torch.manual_seed(1000 + step)
        draws, orig = _record_uniform()
        try:
            out = quiet(model, inp)
        finally:
            torch.Tensor.uniform_ = orig
        lo = loss_fn(out, {"rgb": T(rgb_gt)})
        opt.zero_grad()
        lo["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        arrays[f"s{step}:n_draws"] = np.int64(len(draws))
        for i, dv in enumerate(draws):
            arrays[f"s{step}:draw{i}"] = dv
        for k in ("loss", "rgb_loss", "eikonal_loss", "mask_loss"):
            arrays[f"s{step}:{k}"] = np.float64(lo[k].item())
        arrays[f"s{step}:total_grad_norm"] = np.float64(gn.item())
        arrays[f"s{step}:network_object_mask"] = out["network_object_mask"].numpy()
        arrays[f"s{step}:points"] = out["points"].detach().numpy()
        arrays[f"s{step}:rgb_values"] = out["rgb_values"].detach().numpy()
        arrays[f"s{step}:sdf_output"] = out["sdf_output"].detach().numpy()
        arrays[f"s{step}:grad_theta"] = out["grad_theta"].detach().numpy()
        if step == 0:
            for k, p in model.named_parameters():
                if p.grad is None:
                    arrays[f"s0:gradnorm:{k}"] = np.float64(-1.0)
                else:
                    arrays[f"s0:gradnorm:{k}"] = np.float64(p.grad.double().norm().item())
        opt.step()
        if step in (0, n_steps - 1):
            for k, p in model.named_parameters():
                v = p.detach().numpy()
                arrays[f"s{step}:pnorm:{k}"] = np.float64(np.linalg.norm(v.astype(np.float64)))
                idx = _sample_idx(v.shape, 64, 9)
                arrays[f"s{step}:pidx:{k}"] = idx.astype(np.int64)
                arrays[f"s{step}:pval:{k}"] = v.reshape(-1)[idx]
    arrays["param_names"] = np.asarray(names)
    save("idr_step_" + cfg, **arrays)


def gen_idr_step_C2():
    gen_idr_step("C2", 2048, 52, 1, bias=1.0)


def gen_idr_step_C4():
    """BASELINE configs[3] per-GPU shape: T = 2^22 (223.5 MiB of tables), 2048 rays, one iteration."""
    gen_idr_step("C4", 2048, 53, 1, bias=1.0)


NFFB_STEP = {"C3": ("FFB", 4096, 54), "C5": ("StyleModNFFB", 2048, 55)}


def nffb_conf(embed_type):
    c = idr_conf("C1")
    c["implicit_network"]["multires"] = 6
    c["embedding_network"] = dict(embed_type=embed_type, log2_max_hash_size=5, max_points_per_entry=2,
                                  base_resolution=16, desired_resolution=512, bound=1.0)
    return Conf(c)


def gen_idr_step_nffb(tag):
    """One full iteration (forward + IDRLoss + backward + clip + Adam) with a filter-bank embedder:
    BASELINE configs[2] ('FFB', 4096 rays) and configs[4] ('StyleModNFFB', 2048 rays; fp32 - the reference has no
    reduced-precision path)."""
    embed_type, n, seed = NFFB_STEP[tag]
    model = quiet(IDRNetwork, nffb_conf(embed_type))
    E = 3 + 8 + 8 * 6
    sd = model.implicit_network.state_dict()
    for k, v in P.make_sdf_params(seed + 7, E, (512,) * 8, 257, (4,), 1.0, 0.1, 0.1).items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = T(v)
    for k, v in P.make_nffb_params(seed + 3, 6, embed_type == "StyleModNFFB", 0.3).items():
        kk = "embed_model.embedder_obj." + k
        assert sd[kk].shape == v.shape, (kk, sd[kk].shape, v.shape)
        sd[kk] = T(v)
    model.implicit_network.load_state_dict(sd)
    vlevels, vB, _, _ = P.make_embedder_state(seed + 20, "viewdir", 0.5)
    load_embedder(model.rendering_network.embed_model.embedder_obj, vlevels, vB)
    sd = model.rendering_network.state_dict()
    for k, v in P.make_render_params(seed + 9, d_in0=sd["lin0.weight_v"].shape[1]).items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = T(v)
    model.rendering_network.load_state_dict(sd)
    cam, dirs = P.make_rays(seed + 50, n)
    z = -cam[0] / np.linalg.norm(cam[0])
    xax = np.cross(np.array([0.0, 1.0, 0.0]), z)
    xax /= np.linalg.norm(xax)
    yax = np.cross(z, xax)
    R = np.stack([xax, yax, z], 1)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = cam[0]
    dc = dirs[0].astype(np.float64) @ R
    uv = (dc[:, :2] / dc[:, 2:3]).astype(np.float32).reshape(1, n, 2)
    intr = np.eye(4, dtype=np.float32).reshape(1, 4, 4)
    rs = np.random.RandomState(seed + 60)
    object_mask = (rs.uniform(0, 1, n) < 0.85).reshape(1, n)
    rgb_gt = rs.uniform(-1, 1, (1, n, 3)).astype(np.float32)
    inp = dict(intrinsics=T(intr), uv=T(uv), pose=T(pose.reshape(1, 4, 4)), object_mask=T(object_mask))
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    model.train()
    torch.manual_seed(1000)
    draws, orig = _record_uniform()
    try:
        out = quiet(model, inp)
    finally:
        torch.Tensor.uniform_ = orig
    lo = loss_fn(out, {"rgb": T(rgb_gt)})
    model.zero_grad()
    lo["loss"].backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    arrays = dict(intrinsics=intr, uv=uv, pose=pose.reshape(1, 4, 4), object_mask=object_mask, rgb_gt=rgb_gt,
                  seed=np.int64(seed), embed_type=np.asarray(embed_type))
    arrays["s0:n_draws"] = np.int64(len(draws))
    for i, dv in enumerate(draws):
        arrays[f"s0:draw{i}"] = dv
    for k in ("loss", "rgb_loss", "eikonal_loss", "mask_loss"):
        arrays[f"s0:{k}"] = np.float64(lo[k].item())
    arrays["s0:total_grad_norm"] = np.float64(gn.item())
    arrays["s0:network_object_mask"] = out["network_object_mask"].numpy()
    arrays["s0:rgb_values"] = out["rgb_values"].detach().numpy()
    arrays["s0:sdf_output"] = out["sdf_output"].detach().numpy()
    arrays["s0:grad_theta"] = out["grad_theta"].detach().numpy()
    for k, p_ in model.named_parameters():
        arrays[f"s0:gradnorm:{k}"] = np.float64(-1.0 if p_.grad is None else p_.grad.double().norm().item())
    arrays["param_names"] = np.asarray([k for k, _ in model.named_parameters()])
    save("idr_step_" + tag, **arrays)


def gen_init_rng():
    """Seed-for-seed initialisation parity: checksums of the reference's freshly constructed params."""
    arrays = {}
    for cfg in ("C1", "shipped"):
        torch.manual_seed(1234)
        model = quiet(IDRNetwork, idr_conf(cfg))
        for k, v in model.state_dict().items():
            a = v.numpy().astype(np.float64)
            arrays[f"{cfg}:{k}:sum"] = np.float64(a.sum())
            arrays[f"{cfg}:{k}:abs"] = np.float64(np.abs(a).sum())
            arrays[f"{cfg}:{k}:head"] = v.numpy().reshape(-1)[:8].copy()
        arrays[f"{cfg}:names"] = np.asarray(list(model.state_dict().keys()))
    save("init_rng", **arrays)


def gen_camera():
    rs = np.random.RandomState(71)
    n = 64
    uv = rs.uniform(0, 200, (2, n, 2)).astype(np.float32)
    K = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    K[:, 0, 0] = [150.0, 210.0]
    K[:, 1, 1] = [160.0, 205.0]
    K[:, 0, 2] = [100.0, 98.0]
    K[:, 1, 2] = [90.0, 101.0]
    K[:, 0, 1] = [0.0, 0.5]
    q = rs.standard_normal((2, 4)).astype(np.float32)
    t = rs.standard_normal((2, 3)).astype(np.float32)
    pose7 = np.concatenate([q, t], 1)
    d7, c7 = ref_rend.get_camera_params(T(uv), T(pose7), T(K))
    Rm = ref_rend.quat_to_rot(T(q)).numpy()
    pose44 = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    pose44[:, :3, :3] = Rm
    pose44[:, :3, 3] = t
    d4, c4 = ref_rend.get_camera_params(T(uv), T(pose44), T(K))
    save("camera", uv=uv, K=K, pose7=pose7, pose44=pose44, dirs7=d7.numpy(), cam7=c7.numpy(),
         dirs44=d4.numpy(), cam44=c4.numpy(), R=Rm)


def gen_nffb():
    """FourierFilterBanks ('FFB') and its style-modulated variant ('StyleModNFFB'): embedder outputs and
    ImplicitNetwork forward / gradient / parameter-gradient norms with the reference's own (seeded) init."""
    from model.custom_embedder_decoder import Custom_Embedding_Network
    for tag, et in (("ffb", "FFB"), ("stylemod", "StyleModNFFB")):
        for L in (6, 8):
            torch.manual_seed(77)
            emb = quiet(Custom_Embedding_Network, 3, [3, 512], et, L, 5, 2, 16, 512, 1.0)
            sd = {k: v.clone() for k, v in emb.state_dict().items()}
            # make the hash features matter
            for k in sd:
                if "embedding.weight" in k:
                    sd[k] = torch.from_numpy(np.random.RandomState(5).uniform(-0.5, 0.5, tuple(sd[k].shape)).astype(np.float32))
            emb.load_state_dict(sd)
            x = P.make_points(300 + L, 256, -1.0, 1.0)
            xt = T(x.copy()).requires_grad_(True)
            y = emb(xt)
            R = np.random.RandomState(9).standard_normal(tuple(y.shape)).astype(np.float32)
            (y * T(R)).sum().backward()
            arrays = dict(x=x, out=y.detach().numpy(), R=R, dx=xt.grad.numpy(), L=np.int64(L))
            for k, v in sd.items():
                arrays["sd:" + k] = v.numpy()
            for k, p in emb.named_parameters():
                arrays["gn:" + k] = np.float64(0.0 if p.grad is None else p.grad.double().norm().item())
            save(f"nffb_{tag}_L{L}", **arrays)
    # full SDF network on top of the FFB embedder (config 3 shape, narrow MLP to keep the fixture small)
    for tag, et in (("ffb", "FFB"), ("stylemod", "StyleModNFFB")):
        torch.manual_seed(78)
        net = quiet(ImplicitNetwork, 16, 3, 1, [128] * 8, True, 0.6, [4], True, multires=6, embed_type=et,
                    log2_max_hash_size=5, max_points_per_entry=2, base_resolution=16, desired_resolution=512, bound=1.0)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        rs = np.random.RandomState(6)
        for k in sd:
            if "embedding.weight" in k:
                sd[k] = torch.from_numpy(rs.uniform(-0.5, 0.5, tuple(sd[k].shape)).astype(np.float32))
        sd["lin0.weight_v"][:, 3:] = torch.from_numpy(rs.normal(0, 0.05, tuple(sd["lin0.weight_v"][:, 3:].shape)).astype(np.float32))
        net.load_state_dict(sd)
        x = P.make_points(400, 192, -1.0, 1.0)
        net.eval()
        with torch.no_grad():
            out = net(T(x.copy())).numpy()
        net.train()
        g = net.gradient(T(x.copy()))
        eik = ((g[:, 0, :].norm(2, dim=1) - 1) ** 2).mean()
        net.zero_grad()
        eik.backward()
        arrays = dict(x=x, out=out, gradient=g.detach().numpy()[:, 0], eik=np.float64(eik.item()))
        for k, v in sd.items():
            arrays["sd:" + k] = v.numpy()
        for k, p in net.named_parameters():
            arrays["g2n:" + k] = np.float64(-1.0 if p.grad is None else p.grad.double().norm().item())
        save(f"nffb_sdf_{tag}", **arrays)


def gen_idr_eval():
    """IDRNetwork.forward in eval mode (the evaluation/eval.py caller: model.eval(); model(input))."""
    seed = 61
    cfg = "C1"
    model = quiet(IDRNetwork, idr_conf(cfg))
    L = P.CONFIGS[cfg][0]
    levels, B, _, _ = P.make_embedder_state(seed, cfg, 0.05)
    load_embedder(model.implicit_network.embed_model.embedder_obj, levels, B)
    sd = model.implicit_network.state_dict()
    for k, v in P.make_sdf_params(seed + 7, 3 + 4 * L, (512,) * 8, 257, (4,), 0.6, 0.1, 0.1).items():
        sd[k] = T(v)
    model.implicit_network.load_state_dict(sd)
    vlevels, vB, _, _ = P.make_embedder_state(seed + 20, "viewdir", 0.5)
    load_embedder(model.rendering_network.embed_model.embedder_obj, vlevels, vB)
    sd = model.rendering_network.state_dict()
    for k, v in P.make_render_params(seed + 9).items():
        sd[k] = T(v)
    model.rendering_network.load_state_dict(sd)
    n = 256
    cam, dirs = P.make_rays(seed + 50, n)
    z = -cam[0] / np.linalg.norm(cam[0])
    xax = np.cross(np.array([0.0, 1.0, 0.0]), z)
    xax /= np.linalg.norm(xax)
    yax = np.cross(z, xax)
    R = np.stack([xax, yax, z], 1)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = cam[0]
    dc = dirs[0].astype(np.float64) @ R
    uv = (dc[:, :2] / dc[:, 2:3]).astype(np.float32).reshape(1, n, 2)
    intr = np.eye(4, dtype=np.float32).reshape(1, 4, 4)
    object_mask = (np.random.RandomState(seed).uniform(0, 1, n) < 0.85).reshape(1, n)
    inp = dict(intrinsics=T(intr), uv=T(uv), pose=T(pose.reshape(1, 4, 4)), object_mask=T(object_mask))
    model.eval()
    out = quiet(model, inp)
    save("idr_eval_C1", intrinsics=intr, uv=uv, pose=pose.reshape(1, 4, 4), object_mask=object_mask,
         seed=np.int64(seed), points=out["points"].detach().numpy(), rgb_values=out["rgb_values"].detach().numpy(),
         sdf_output=out["sdf_output"].detach().numpy(), network_object_mask=out["network_object_mask"].numpy())


GENS = dict(idr_step_C4=gen_idr_step_C4, idr_step_C3=lambda: gen_idr_step_nffb("C3"),
            idr_step_C5=lambda: gen_idr_step_nffb("C5"), sdf_C2=lambda: gen_sdf(("C2",)), raytrace_C2=lambda: gen_raytrace(("C2",)), idr_step_C2=gen_idr_step_C2,
            idr_eval=gen_idr_eval, nffb=gen_nffb, levels=gen_levels, hash_ids=gen_hash_ids, encode=gen_encode, encode_bwd=gen_encode_bwd,
            sdf=gen_sdf, raytrace=gen_raytrace, idr_step=gen_idr_step, init_rng=gen_init_rng,
            camera=gen_camera)

if __name__ == "__main__":
    which = sys.argv[1:] or [k for k in GENS if k not in ("sdf_C2", "raytrace_C2")]   # (subsets of sdf / raytrace)
    for w in which:
        print(f"== {w}")
        GENS[w]()
