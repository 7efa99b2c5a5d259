"""Pins oracle/ (the CPU restatement) against the fixtures generated from the reference
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

import params as P
from oracle import c_oracle as O

# SURVEY.md Appendix A known-answer values (reference hash_func + bin_mask order)
KAT = [
    (16, 4096, (0.25, -0.5, 0.75), (4, -8, 12), [2976, 2977, 2979, 2978, 529, 528, 530, 531]),
    (16, 4096, (0, 0, 0), (0, 0, 0), [0, 1, 3, 2, 2481, 2480, 2482, 2483]),
    (16, 4096, (-1, -1, -1), (-16, -16, -16), [1232, 1233, 1235, 1234, 3713, 3712, 3714, 3715]),
    (101, 524288, (0.25, -0.5, 0.75), (25, -50, 75),
     [481704, 481707, 481711, 481708, 516095, 516092, 516088, 516091]),
    (101, 524288, (-0.999, 0.001, 0.5), (-100, 0, 50),
     [146190, 146191, 146189, 146188, 180703, 180702, 180700, 180701]),
    (512, 524288, (1, 1, 1), (512, 512, 512), [222720, 222721, 222723, 222722, 188337, 188336, 188338, 188339]),
    (512, 4194304, (0.25, -0.5, 0.75), (128, -256, 384),
     [619520, 619521, 619523, 619522, 1179313, 1179312, 1179314, 1179315]),
    (512, 4194304, (-1, -1, -1), (-512, -512, -512),
     [825856, 825857, 825859, 825858, 267185, 267184, 267186, 267187]),
]


@pytest.mark.parametrize("res,rows,x,xi,ids", KAT)
def test_known_answers(res, rows, x, xi, ids):
    gxi, gids = O.corner_ids(np.asarray([x], np.float32), res, rows)
    assert gxi[0].tolist() == list(xi)
    assert gids[0].tolist() == ids


@pytest.mark.parametrize("cfg", list(P.CONFIGS))
def test_level_table(golden, cfg):
    g = golden("levels")
    L, T, b, d = P.CONFIGS[cfg]
    grid = O.Grid(L, T, b, d)
    assert grid.res.tolist() == g[cfg + "_res"].tolist()
    assert grid.rows.tolist() == g[cfg + "_rows"].tolist()
    assert grid.E == int(g[cfg + "_E"])
    res, rows = P.level_table(L, T, b, d)
    assert res == g[cfg + "_res"].tolist() and rows == g[cfg + "_rows"].tolist()


def test_hash_ids_bit_exact(golden):
    g = golden("hash_ids")
    for i, (res, rows) in enumerate(g["combos"]):
        xi, ids = O.corner_ids(g[f"x_{i}"], res, rows)
        assert np.array_equal(xi, g[f"xi_{i}"]), (res, rows)
        assert np.array_equal(ids, g[f"ids_{i}"]), (res, rows)


@pytest.mark.parametrize("cfg", ["C1", "C2", "shipped", "viewdir", "tiny"])
def test_encode_fwd(golden, cfg):
    g = golden(f"encode_{cfg}")
    L, T, b, d = P.CONFIGS[cfg]
    grid = O.Grid(L, T, b, d)
    levels, B, _, _ = P.make_embedder_state(int(g["seed"]), cfg, float(g["table_scale"]))
    table = np.concatenate(levels, 0)
    out = O.encode_fwd(grid, g["x"], table, B, 0)
    ref = g["out"]
    nf = 3 + 2 * L
    assert np.array_equal(out[:, :3], ref[:, :3])
    assert np.array_equal(out[:, nf:], ref[:, nf:]), "hash features must be exact in reference mode"
    np.testing.assert_allclose(out[:, 3:nf], ref[:, 3:nf], atol=1e-6, rtol=0)


@pytest.mark.parametrize("cfg", ["C1", "tiny"])
def test_encode_bwd_table(golden, cfg):
    g = golden(f"encode_bwd_{cfg}")
    L, T, b, d = P.CONFIGS[cfg]
    grid = O.Grid(L, T, b, d)
    assert grid.total_rows == int(g["total_rows"])
    dt = O.encode_bwd_table(grid, g["x"], g["d_out"], 0)
    nz = np.nonzero(np.abs(dt).sum(1))[0]
    assert np.array_equal(nz, g["nz_rows"])
    np.testing.assert_allclose(dt[nz], g["nz_grad"], rtol=1e-5, atol=1e-6)


def _sdf_oracle(g, cfg, hidden, fvs):
    L, T, b, d = P.CONFIGS[cfg]
    seed = int(g["seed"])
    grid = O.Grid(L, T, b, d)
    levels, B, _, _ = P.make_embedder_state(seed, cfg, float(g["table_scale"]))
    prm = P.make_sdf_params(seed + 7, grid.E, hidden, 1 + fvs, (4,), 0.6, float(g["perturb"]), 0.1)
    return O.SdfOracle(grid, np.concatenate(levels, 0), B, prm)


@pytest.mark.parametrize("tag,cfg", [("full", "C1"), ("init", "C1"), ("narrow", "tiny"), ("C2", "C2")])
def test_sdf_forward(golden, tag, cfg):
    g = golden(f"sdf_{tag}")
    net = _sdf_oracle(g, cfg, tuple(g["hidden"].tolist()), int(g["fvs"]))
    out = net(g["x"])
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=2e-6)


def test_sphere_intersection(golden):
    g = golden("raytrace_init")
    t2, m = O.sphere_intersection(g["cam_loc"], g["ray_dirs"], 1.0)
    assert np.array_equal(m, g["mask_intersect"].reshape(-1))
    np.testing.assert_allclose(t2, g["sphere_intersections"].reshape(-1, 2), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag,cfg", [("init", "C1"), ("bumpy", "C1"), ("C2", "C2")])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_ray_ref_golden(golden, tag, cfg, mode):
    """oracle/ray_ref.py (the ray search restated independently of the product) against the reference's own
    RayTracing.forward outputs: masks exact, distances / points to fp32 round-off, same number of SDF evaluations.
    The SDF callable is oracle/torch_ref.RefImplicit on the CPU (torch sgemm, like the reference run)."""
    import torch
    from helpers import make_implicit
    from oracle import ray_ref, torch_ref as R
    g = golden(f"raytrace_{tag}")
    net = make_implicit(cfg, (512,) * 8, 256, int(g["seed"]), float(g["perturb"]), float(g["table_scale"]),
                        device="cpu", bias=float(g["bias"]) if "bias" in g.files else 0.6)
    sdf_net = R.RefImplicit(R._grid_from(net.embed_model.embedder_obj), R._lins(net), net.skip_in)
    torch.set_num_threads(8)
    rt = ray_ref.RayTraceRef(1.0, 5.0e-5, 0.5, 3, 10, 100, 8)
    rt.training = mode == "train"
    rt.steps = torch.from_numpy(g["steps"])
    with torch.no_grad():
        pts, mask, t = rt(sdf_net.sdf, torch.from_numpy(g["cam_loc"]), torch.from_numpy(g["object_mask"]),
                          torch.from_numpy(g["ray_dirs"]))
    flips = int((mask.numpy() != g[f"{mode}_mask"]).sum())
    assert flips <= 1, flips                      # (one ray on a threshold may flip with the BLAS thread count)
    same = mask.numpy() == g[f"{mode}_mask"]
    np.testing.assert_allclose(t.numpy()[same], g[f"{mode}_dists"][same], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(pts.numpy()[same], g[f"{mode}_points"][same], rtol=2e-5, atol=5e-6)
    assert abs(rt.sdf_evals - int(g[f"{mode}_sdf_evals"])) <= 0.01 * int(g[f"{mode}_sdf_evals"])
