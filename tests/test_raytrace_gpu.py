"""GPU parity of RayTracing.forward (plugin point #3) against reference outputs for the same SDF
weights, rays, object mask and injected random step fractions."""
import numpy as np
import pytest
import torch

from helpers import make_implicit

pytestmark = pytest.mark.gpu


def _cfg(tag):
    return "C2" if tag == "C2" else "C1"


def _net(g, tag):
    return make_implicit(_cfg(tag), (512,) * 8, 256, int(g["seed"]), float(g["perturb"]), float(g["table_scale"]),
                         bias=float(g["bias"]) if "bias" in g.files else 0.6)


def _run(g, mode, tag="init"):
    from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing
    net = _net(g, tag)
    net.eval()
    rt = RayTracing(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=3,
                    sphere_tracing_iters=10, n_steps=100, n_secant_steps=8).cuda()
    rt.train(mode == "train")
    rt.steps_override = torch.from_numpy(g["steps"])
    evals = []

    def sdf(p):
        evals.append(p.shape[0])
        return net.sdf(p)

    with torch.no_grad():
        pts, mask, dists = rt(sdf=sdf, cam_loc=torch.from_numpy(g["cam_loc"]).cuda(),
                              object_mask=torch.from_numpy(g["object_mask"]).cuda(),
                              ray_directions=torch.from_numpy(g["ray_dirs"]).cuda())
    return pts.cpu().numpy(), mask.cpu().numpy(), dists.cpu().numpy(), sum(evals)


@pytest.mark.parametrize("tag", ["init", "bumpy", "C2"])   # C2: the benchmarked configuration, 2048 rays
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_raytracing_golden(golden, tag, mode):
    g = golden(f"raytrace_{tag}")
    pts, mask, dists, n_evals = _run(g, mode, tag)
    ref_mask, ref_d, ref_p = g[f"{mode}_mask"], g[f"{mode}_dists"], g[f"{mode}_points"]
    # The SDF values differ from the reference's CPU GEMM in the last bits, and every decision of the
    # tracer is a threshold on them (sdf > 5e-5, sign changes), so a ray sitting exactly on a
    # threshold may legitimately take the other branch: allow <= 1 % such rays, everything else tight.
    mism = mask != ref_mask
    print(f"raytrace {tag}/{mode}: {mism.sum()} / {mism.size} mask mismatches, {n_evals} SDF evaluations "
          f"(reference {int(g[f'{mode}_sdf_evals'])}), hits {int(mask.sum())}")
    assert mism.mean() <= 0.01, f"{mism.sum()} mask mismatches"
    same = ~mism
    dd = np.abs(dists - ref_d)[same]
    tol = 1e-5 * np.maximum(1.0, np.abs(ref_d[same]))
    loose = dd > 20 * tol
    assert loose.mean() <= 0.02, f"{loose.sum()} rays with dist error > 2e-4: max {dd.max()}"
    assert np.median(dd) <= 1e-6
    pd = np.abs(pts - ref_p).max(1)[same]
    assert (pd > 1e-3).mean() <= 0.02
    # same amount of SDF work as the reference (within the few rays that flipped a branch)
    assert abs(n_evals - int(g[f"{mode}_sdf_evals"])) <= 0.03 * int(g[f"{mode}_sdf_evals"])


def test_sphere_intersection_golden(golden):
    from hashmodnffbanks_idr_amd.utils import rend_util
    g = golden("raytrace_init")
    t, m = rend_util.get_sphere_intersection(torch.from_numpy(g["cam_loc"]).cuda(),
                                             torch.from_numpy(g["ray_dirs"]).cuda(), r=1.0)
    assert np.array_equal(m.cpu().numpy(), g["mask_intersect"])
    np.testing.assert_allclose(t.cpu().numpy(), g["sphere_intersections"], rtol=1e-6, atol=1e-6)


def _device_vs_host(tag, golden, mode, n_rays=None, seed=0, tile=64):
    """The sync-free device tracer must reproduce the generic (torch op) tracer when both use the
    same SDF kernel tile size (so the SDF values are bit-identical)."""
    from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing
    import params as P
    g = golden(f"raytrace_{tag}")
    net = _net(g, tag)
    net.eval()
    net.sdf_tile_points = tile
    if n_rays is None:
        cam, dirs, om = g["cam_loc"], g["ray_dirs"], g["object_mask"]
    else:
        cam, dirs = P.make_rays(seed, n_rays)
        om = np.random.RandomState(seed).uniform(0, 1, n_rays) < 0.7
    outs = []
    for dev_tracer in (True, False):
        rt = RayTracing(1.0, 5.0e-5, 0.5, 3, 10, 100, 8).cuda()
        rt.train(mode == "train")
        rt.use_device_tracer = dev_tracer
        rt.steps_override = torch.from_numpy(g["steps"])
        with torch.no_grad():
            outs.append(rt(sdf=net.sdf, cam_loc=torch.from_numpy(cam).cuda(), object_mask=torch.from_numpy(om).cuda(),
                           ray_directions=torch.from_numpy(dirs).cuda()))
        if dev_tracer:
            st = rt.last_stats
            assert st["unfinished"] == 0
    (p1, m1, d1), (p2, m2, d2) = outs
    if tile == 0:
        # tile size left to the library: the device tracer's persistent march picks the small-tile body from each
        # WORKGROUP's pending points (8 rays), the generic tracer's SDF calls pick it from the whole batch's - the bodies
        # (4 / 8 / 16 points: different MFMA shapes) agree to ~1e-7, not to the bit, so a ray at a threshold may take
        # another branch.  Bit-for-bit equality is asserted with fixed tile sizes (16: the persistent kernel, 64).
        flips = int((m1 != m2).sum())
        both = m1 & m2
        dd = (d1 - d2).abs()
        close = dd <= 1e-4 * (1.0 + d2.abs())
        frac_far = 1.0 - float(close.float().mean())
        print(f"tile 0: {flips} mask flips of {m1.numel()} rays, max |d dist| on common hits "
              f"{float(dd[both].max()) if bool(both.any()) else 0.0:.3e}, rays beyond 1e-4: {frac_far:.4f}")
        assert flips <= max(1, m1.numel() // 200)
        assert frac_far <= 0.01
        return st
    assert torch.equal(m1, m2)
    assert torch.equal(d1, d2), (d1 - d2).abs().max()
    assert torch.equal(p1, p2), (p1 - p2).abs().max()
    return st


@pytest.mark.parametrize("tag", ["init", "bumpy"])
@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("tile", [16, 64])
def test_device_tracer_equals_generic_tracer(golden, tag, mode, tile):
    _device_vs_host(tag, golden, mode, tile=tile)


def test_device_tracer_equals_generic_tracer_2048_rays(golden):
    st = _device_vs_host("bumpy", golden, "train", n_rays=2048, seed=11, tile=16)
    assert st["sdf_evals"] > 2048 * 20


@pytest.mark.parametrize("tile", [0, 16, 64])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_device_tracer_equals_generic_tracer_C2(golden, mode, tile):
    """Benchmarked configuration (L=16, T=2^19 -> E=67), 2048 rays.  tile 0 = the tile size is chosen per call from the
    live point count (see _device_vs_host: near-equality).  tile 16 runs the march as ONE persistent launch with every
    round on the 16-point body (hm_sdf.hip: trace_march_kernel), tile 64 the launch-per-round form: both must agree
    with the generic tracer bit for bit."""
    st = _device_vs_host("C2", golden, mode, tile=tile)
    assert st["sdf_evals"] > 2048 * 10 and st.get("nonfinite", 0) == 0


@pytest.mark.parametrize("tile", [0, 16, 64])
def test_device_tracer_equals_generic_tracer_trilinear(golden, tile):
    """frac_mode = 'trilinear' (the opt-in interpolating encoder): the device tracer - its TRILINEAR kernel instantiations,
    with tile 0 the one-launch scan + secant form and the filler tiles - against the generic tracer on the same ops,
    training mode, 2048 rays: bit for bit at fixed tile sizes, the tile-0 criterion of _device_vs_host otherwise."""
    from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing
    import params as P
    g = golden("raytrace_bumpy")
    net = _net(g, "bumpy")
    net.embed_model.embedder_obj.frac_mode = "trilinear"
    net.eval()
    net.sdf_tile_points = tile
    cam, dirs = P.make_rays(17, 2048)
    om = np.random.RandomState(17).uniform(0, 1, 2048) < 0.7
    outs = []
    for dev_tracer in (True, False):
        rt = RayTracing(1.0, 5.0e-5, 0.5, 3, 10, 100, 8).cuda()
        rt.train(True)
        rt.use_device_tracer = dev_tracer
        rt.steps_override = torch.from_numpy(g["steps"])
        with torch.no_grad():
            outs.append(rt(sdf=net.sdf, cam_loc=torch.from_numpy(cam).cuda(), object_mask=torch.from_numpy(om).cuda(),
                           ray_directions=torch.from_numpy(dirs).cuda()))
        if dev_tracer:
            assert rt.last_stats["unfinished"] == 0 and rt.last_stats.get("nonfinite", 0) == 0
            assert rt.last_stats["mask_loss_rays"] * 100 > 8192
    (p1, m1, d1), (p2, m2, d2) = outs
    if tile == 0:
        assert int((m1 != m2).sum()) <= 10
        rel = ((d1 - d2).abs() / (1.0 + d2.abs()))
        assert float((rel > 1e-4).float().mean()) <= 0.01
    else:
        assert torch.equal(m1, m2) and torch.equal(d1, d2) and torch.equal(p1, p2)


@pytest.mark.parametrize("tag", ["bumpy", "C2"])
@pytest.mark.parametrize("head", [0, 16])
def test_scan_and_secant_in_one_launch(golden, tag, head, monkeypatch):
    """Training searches with the tile size left to the library run the closest-approach scan and the secant refinement
    as ONE launch (hm_sdf.hip: sdf_scan_secant_kernel - workgroups without secant rays start on the scan at once, its
    64-point tiles are handed out by an atomic cursor); HM_TRACE_OVERLAP=0 is the form before it (the scan inside the
    sampler's launch, the secant behind it).  The scan's values do not depend on who evaluates which tile: the
    closest-approach results must be bit-identical.  The secant runs on 16-point tiles under a long scan and on 4 / 8-point
    ones alone (different MFMA shapes, ~1e-7 apart; the secant steps amplify that on a bumpy surface: 2e-5 seen): its
    rays agree to 1e-4 relative, the network mask may flip for a ray at the threshold.  Both sampler forms (head = 0: single pass, 16: lazy)."""
    from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing
    import params as P
    g = golden(f"raytrace_{tag}")
    net = _net(g, tag)
    net.eval()
    net.sdf_tile_points = 0
    n_rays = 2048
    cam, dirs = P.make_rays(21, n_rays)
    om = np.random.RandomState(21).uniform(0, 1, n_rays) < 0.6
    outs, stats = [], []
    for ov in ("1", "0"):
        monkeypatch.setenv("HM_TRACE_OVERLAP", ov)
        rt = RayTracing(1.0, 5.0e-5, 0.5, 3, 10, 100, 8).cuda()
        rt.train(True)
        rt.sampler_head = head
        rt.steps_override = torch.from_numpy(g["steps"])
        with torch.no_grad():
            outs.append(rt(sdf=net.sdf, cam_loc=torch.from_numpy(cam).cuda(), object_mask=torch.from_numpy(om).cuda(),
                           ray_directions=torch.from_numpy(dirs).cuda()))
        stats.append(dict(rt.last_stats))
    (p1, m1, d1), (p2, m2, d2) = outs
    s1, s2 = stats
    print(f"{tag} head {head}: stats {s1}")
    assert s1["unfinished"] == 0 and s1.get("nonfinite", 0) == 0
    assert s1["sdf_evals"] == s2["sdf_evals"] and s1["mask_loss_rays"] == s2["mask_loss_rays"] > 80   # (> 8192 scan points)
    flips = int((m1 != m2).sum())
    assert flips <= 2
    miss = ~m1 & ~m2
    assert torch.equal(d1[miss], d2[miss]) and torch.equal(p1[miss], p2[miss])      # closest approach / sphere misses
    hit = m1 & m2
    dd = (d1 - d2)[hit].abs()
    print(f"  {int(hit.sum())} common hits, max |d dist| {float(dd.max()):.3e}, bit-equal {float((dd == 0).float().mean()):.3f}, "
          f"{flips} mask flips")
    assert float((dd / (1.0 + d2[hit].abs())).max()) <= 1e-4      # (the tile-0 criterion of _device_vs_host)


@pytest.mark.parametrize("n_rays", [77, 300, 2048])
@pytest.mark.parametrize("case", ["plain", "all_miss_sphere", "mask_all_false", "mask_all_true"])
def test_scan_and_secant_in_one_launch_edge_cases(golden, case, n_rays, monkeypatch):
    """The one-launch form (see test_scan_and_secant_in_one_launch) where one of its parts is empty or small: no secant
    ray at all (every ray misses the sphere), every ray a mask-loss ray (the scan is the whole coarse work, the sampler's
    launch has no point of its own and takes no filler tile), no mask-loss ray (empty scan), 77 rays (all launches on the
    small-tile kernels) and 300 (scan above, sampler possibly below the 64-point kernel's range)."""
    from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing
    import params as P
    g = golden("raytrace_bumpy")
    net = _net(g, "bumpy")
    net.eval()
    net.sdf_tile_points = 0
    cam, dirs = P.make_rays(31, n_rays)
    om = np.random.RandomState(31).uniform(0, 1, n_rays) < 0.6
    if case == "all_miss_sphere":
        dirs = -dirs
    if case == "mask_all_false":
        om[:] = False
    if case == "mask_all_true":
        om[:] = True
    outs, stats = [], []
    for ov in ("1", "0"):
        monkeypatch.setenv("HM_TRACE_OVERLAP", ov)
        rt = RayTracing(1.0, 5.0e-5, 0.5, 3, 10, 100, 8).cuda()
        rt.train(True)
        rt.steps_override = torch.from_numpy(g["steps"])
        with torch.no_grad():
            outs.append(rt(sdf=net.sdf, cam_loc=torch.from_numpy(cam).cuda(), object_mask=torch.from_numpy(om).cuda(),
                           ray_directions=torch.from_numpy(np.ascontiguousarray(dirs)).cuda()))
        stats.append(dict(rt.last_stats))
    (p1, m1, d1), (p2, m2, d2) = outs
    s1, s2 = stats
    assert s1["unfinished"] == 0 and s1.get("nonfinite", 0) == 0 and s1["sdf_evals"] == s2["sdf_evals"]
    assert int((m1 != m2).sum()) <= 1
    both = (m1 == m2)
    rel = ((d1 - d2).abs() / (1.0 + d2.abs()))[both]
    assert rel.numel() == 0 or float(rel.max()) <= 1e-4
    if s1["mask_loss_rays"] * 100 > 8192 and (s1["sampler_points"] + s1["mask_loss_rays"] * 100) > 8192:
        miss = ~m1 & ~m2          # both forms scan on the 64-point kernel: bit-identical closest-approach results
        assert torch.equal(d1[miss], d2[miss]) and torch.equal(p1[miss], p2[miss])
    if case == "all_miss_sphere":
        assert not bool(m1.any()) and s1["secant_rays"] == 0


@pytest.mark.parametrize("tag,n_rays", [("bumpy", None), ("bumpy", 2048), ("C2", None)])
def test_persistent_march_tail_carries_whole_marches(golden, tag, n_rays, monkeypatch):
    """hm_sdf.hip: trace_march_tail_kernel normally takes over after the 1 + sphere_tracing_iters guaranteed rounds and
    finds work only when a ray ran line searches.  HM_TRACE_TAIL_FIRST=1 hands over after the FIRST round: every ray's
    whole march (all iterations, line searches included) then runs inside the persistent kernel, eight rays per
    workgroup - and must still reproduce the generic tracer bit for bit (16-point tiles on both sides)."""
    monkeypatch.setenv("HM_TRACE_TAIL_FIRST", "1")
    st = _device_vs_host(tag, golden, "train", n_rays=n_rays, seed=3, tile=16)
    assert st["sdf_evals"] > 0 and st.get("nonfinite", 0) == 0
    monkeypatch.setenv("HM_TRACE_TAIL_FIRST", "4")
    _device_vs_host(tag, golden, "eval", n_rays=n_rays, seed=4, tile=16)


@pytest.mark.parametrize("tile", [16, 64])
@pytest.mark.parametrize("case", ["one_ray", "all_miss_sphere", "mask_all_false", "mask_all_true", "odd_count"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_device_tracer_edge_cases(golden, case, mode, tile):
    """Degenerate batches the reference's masked code paths handle implicitly (ray_tracing.py:46-95: empty
    selections, rays that miss the bounding sphere, no / all rays inside the object mask): device tracer ==
    generic tracer, bit for bit."""
    from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing
    import params as P
    g = golden("raytrace_bumpy")
    net = make_implicit("C1", (512,) * 8, 256, int(g["seed"]), float(g["perturb"]), float(g["table_scale"]))
    net.eval()
    net.sdf_tile_points = tile      # 16: persistent march kernel (77 rays: a last workgroup with 5 rays), 64: a launch per round
    n = {"one_ray": 1, "odd_count": 77}.get(case, 64)
    cam, dirs = P.make_rays(5, n)
    om = np.random.RandomState(5).uniform(0, 1, n) < 0.6
    if case == "all_miss_sphere":
        dirs = -dirs                      # every ray points away from the unit sphere
    if case == "mask_all_false":
        om[:] = False
    if case == "mask_all_true":
        om[:] = True
    outs = []
    for dev_tracer in (True, False):
        rt = RayTracing(1.0, 5.0e-5, 0.5, 3, 10, 100, 8).cuda()
        rt.train(mode == "train")
        rt.use_device_tracer = dev_tracer
        rt.steps_override = torch.from_numpy(g["steps"])
        with torch.no_grad():
            outs.append(rt(sdf=net.sdf, cam_loc=torch.from_numpy(cam).cuda(), object_mask=torch.from_numpy(om).cuda(),
                           ray_directions=torch.from_numpy(np.ascontiguousarray(dirs)).cuda()))
    (p1, m1, d1), (p2, m2, d2) = outs
    assert p1.shape == (n, 3) and m1.shape == (n,) and d1.shape == (n,)
    assert torch.equal(m1, m2)
    assert torch.equal(d1, d2) and torch.equal(p1, p2)
    if case == "all_miss_sphere":
        assert not bool(m1.any())


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("tag,n_rays", [("bumpy", 2048), ("bumpy", 77), ("C2", None)])
def test_lazy_sampler_equals_single_pass(golden, tag, n_rays, mode):
    """hm_trace_cfg.sampler_head: evaluating a sampler ray's samples only up to its first sign change (two passes)
    gives the same points / mask / distances, bit for bit, as evaluating all n_steps samples of every unconverged ray
    like the reference does (ray_tracing.py:189-249) - and fewer SDF evaluations.  Fixed 64-point tiles, so that an SDF
    value does not depend on which launch computed it."""
    from hashmodnffbanks_idr_amd.model.ray_tracing import RayTracing
    import params as P
    g = golden(f"raytrace_{tag}")
    net = _net(g, tag)
    net.eval()
    net.sdf_tile_points = 64
    if n_rays is None:
        cam, dirs, om = g["cam_loc"], g["ray_dirs"], g["object_mask"]
    else:
        cam, dirs = P.make_rays(3, n_rays)
        om = np.random.RandomState(3).uniform(0, 1, n_rays) < 0.7
    res = {}
    for head in (0, 1, 5, 16, 50, 98, 99, 1000):
        rt = RayTracing(1.0, 5.0e-5, 0.5, 3, 10, 100, 8).cuda()
        rt.train(mode == "train")
        rt.sampler_head = head
        rt.steps_override = torch.from_numpy(g["steps"])
        with torch.no_grad():
            out = rt(sdf=net.sdf, cam_loc=torch.from_numpy(cam).cuda(), object_mask=torch.from_numpy(om).cuda(),
                     ray_directions=torch.from_numpy(dirs).cuda())
        st = rt.last_stats
        assert st["unfinished"] == 0 and st["nonfinite"] == 0
        res[head] = (out, st)
    (p0, m0, d0), st0 = res[0]
    assert st0["sampler_points"] == 100 * st0["sampler_rays"]
    for head, ((p, m, d), st) in res.items():
        assert torch.equal(m, m0) and torch.equal(d, d0) and torch.equal(p, p0), head
        assert st["sampler_rays"] == st0["sampler_rays"] and st["secant_rays"] == st0["secant_rays"]
        if 1 <= head <= 98:
            assert st["sampler_points"] == (head + 1) * st["sampler_rays"] + (99 - head) * st["sampler_second_pass_rays"]
            assert st["sdf_evals"] <= st0["sdf_evals"]
        else:       # out of range: single pass
            assert st["sdf_evals"] == st0["sdf_evals"]
    print(f"lazy sampler {tag}/{mode}: sampler rays {st0['sampler_rays']}, SDF evaluations "
          + ", ".join(f"head {h}: {r[1]['sdf_evals']}" for h, r in res.items()))
    if tag == "C2":      # near the geometric initialisation the first sign change comes early
        assert res[16][1]["sdf_evals"] < 0.9 * st0["sdf_evals"]


def test_camera_rays_kernel_matches_rend_util(golden):
    """ops.camera_rays (hm_camera_rays: ONE launch) against rend_util.get_camera_params + get_sphere_intersection (the
    reference's expressions on torch ops, ~38 launches) on the cameras of the reference fixture and on random ones:
    directions and camera centres to an ulp or two, hit masks equal away from the sphere's silhouette, t to 1e-6."""
    from hashmodnffbanks_idr_amd import ops
    from hashmodnffbanks_idr_amd.utils import rend_util
    g = golden("camera")
    cases = [(torch.from_numpy(g["uv"]).cuda(), torch.from_numpy(g["pose44"]).cuda(), torch.from_numpy(g["K"]).cuda())]
    gen = torch.Generator(device="cpu").manual_seed(4)
    for Bn, N in ((1, 2048), (3, 777)):
        uv = (torch.rand((Bn, N, 2), generator=gen) * torch.tensor([1600.0, 1200.0])).cuda()
        K = torch.eye(4).repeat(Bn, 1, 1)
        K[:, 0, 0] = 2800 + 100 * torch.rand(Bn, generator=gen); K[:, 1, 1] = 2850.0; K[:, 0, 2] = 810.0; K[:, 1, 2] = 590.0
        K[:, 0, 1] = 0.3
        centre = torch.nn.functional.normalize(torch.randn((Bn, 3), generator=gen), dim=1) * 4.5
        z = torch.nn.functional.normalize(-centre + 0.05 * torch.randn((Bn, 3), generator=gen), dim=1)   # looks at the object
        xax = torch.nn.functional.normalize(torch.linalg.cross(z, torch.randn((Bn, 3), generator=gen)), dim=1)
        P4 = torch.eye(4).repeat(Bn, 1, 1)
        P4[:, :3, 0], P4[:, :3, 1], P4[:, :3, 2], P4[:, :3, 3] = xax, torch.linalg.cross(z, xax), z, centre
        cases.append((uv, P4.cuda(), K.cuda()))
    for ci, (uv, pose, K) in enumerate(cases):
        d_ref, c_ref = rend_util.get_camera_params(uv, pose, K)
        t_ref, h_ref = rend_util.get_sphere_intersection(c_ref, d_ref, r=1.0)
        d, c, t, h = ops.camera_rays(uv, pose, K, 1.0)
        assert torch.equal(c, c_ref)
        dd = (d - d_ref).abs().max().item()
        same = float((d == d_ref).float().mean())
        print(f"camera_rays: max |d dir| {dd:.2e} ({100 * same:.1f} % of the components bit-equal), "
              f"max |d t| {(t - t_ref).abs().max().item():.2e}, hit flips {int((h != h_ref).sum())} of {h.numel()}")
        assert dd <= 2.5e-7
        assert int((h != h_ref).sum()) <= max(1, h.numel() // 1000)
        ok = h & h_ref
        assert ci == 0 or 0.05 * h.numel() < int(ok.sum()) < h.numel()      # the image holds the sphere's silhouette
        inner = ok & ((t_ref[..., 1] - t_ref[..., 0]) > 0.2)               # sqrt() amplifies an ulp of <d, c> at the silhouette
        assert (t - t_ref)[inner].abs().max().item() <= 5e-5           # (|c| = 4.5: an ulp of d moves dot^2 by 4e-6, root 0.1)
        assert (t - t_ref)[ok].abs().max().item() <= 5e-3
