"""CPU tests of host logic: seed-for-seed initialisation parity with the reference constructor,
camera helpers, state_dict key compatibility of the whole IDRNetwork."""
import numpy as np
import pytest
import torch

from helpers import idr_conf


@pytest.mark.parametrize("cfg", ["C1", "shipped"])
def test_init_rng_parity(golden, cfg):
    """torch.manual_seed(s); IDRNetwork(conf) produces the reference's freshly initialised parameters."""
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    g = golden("init_rng")
    torch.manual_seed(1234)
    model = IDRNetwork(idr_conf(cfg))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    names = [str(n) for n in g[f"{cfg}:names"]]
    assert sorted(sd.keys()) == sorted(names)
    for k in names:
        a = sd[k].numpy().astype(np.float64)
        assert np.array_equal(sd[k].numpy().reshape(-1)[:8], g[f"{cfg}:{k}:head"]), k
        assert abs(a.sum() - float(g[f"{cfg}:{k}:sum"])) <= 1e-9 * max(1.0, abs(float(g[f'{cfg}:{k}:abs']))), k
        assert abs(np.abs(a).sum() - float(g[f"{cfg}:{k}:abs"])) <= 1e-9 * max(1.0, float(g[f"{cfg}:{k}:abs"])), k


def test_camera_params(golden):
    from hashmodnffbanks_idr_amd.utils import rend_util
    g = golden("camera")
    T = torch.from_numpy
    d7, c7 = rend_util.get_camera_params(T(g["uv"]), T(g["pose7"]), T(g["K"]))
    np.testing.assert_allclose(d7.numpy(), g["dirs7"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(c7.numpy(), g["cam7"], rtol=0, atol=0)
    d4, c4 = rend_util.get_camera_params(T(g["uv"]), T(g["pose44"]), T(g["K"]))
    np.testing.assert_allclose(d4.numpy(), g["dirs44"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rend_util.quat_to_rot(T(g["pose7"][:, :4])).numpy(), g["R"], rtol=1e-6, atol=1e-6)


def test_get_class_plugin_point():
    from hashmodnffbanks_idr_amd.utils.general import get_class
    cls = get_class("hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer.IDRNetwork")
    assert cls.__name__ == "IDRNetwork"
    assert get_class("hashmodnffbanks_idr_amd.model.loss.IDRLoss").__name__ == "IDRLoss"


def test_pack_layer_layout():
    """w_packed[((u*n_oct+g)*64+l)*4+s] == W[32u+(l&31)][8g+4(l>>5)+s] (include/hashmod.h contract)."""
    from hashmodnffbanks_idr_amd import ops
    W = torch.arange(45 * 83, dtype=torch.float32).reshape(45, 83)
    Wp, n_tiles, octs, srcs = ops.pack_mlp_layer(W, [(0, 64), (1, 19)])
    assert n_tiles == 2 and octs == [8, 3] and srcs == [0, 1]
    flat = Wp.reshape(-1)
    n_oct = 11
    padded = torch.zeros(64, 88)
    padded[:45, :64] = W[:, :64]
    padded[:45, 64:83] = W[:, 64:]
    for (u, g, l, s) in [(0, 0, 0, 0), (1, 10, 63, 3), (0, 8, 33, 2), (1, 3, 12, 1), (1, 7, 44, 0)]:
        assert flat[((u * n_oct + g) * 64 + l) * 4 + s] == padded[32 * u + (l & 31), 8 * g + 4 * (l >> 5) + s]


def test_grid_uniform_layout():
    """utils/plots.get_grid_uniform: meshgrid('xy') ordering of the reference (plots.py:227-238)."""
    from hashmodnffbanks_idr_amd.utils import plots
    g = plots.get_grid_uniform(5)
    pts = g["grid_points"].numpy()
    lin = np.linspace(-1.0, 1.0, 5).astype(np.float32)
    assert pts.shape == (125, 3)
    # index = iy*25 + ix*5 + iz  (numpy meshgrid default indexing='xy')
    for iy, ix, iz in [(0, 0, 0), (1, 2, 3), (4, 0, 2), (3, 4, 4)]:
        np.testing.assert_array_equal(pts[iy * 25 + ix * 5 + iz], [lin[ix], lin[iy], lin[iz]])
    vol = plots.sdf_volume(lambda p: p[:, 0] + 10 * p[:, 1] + 100 * p[:, 2], g)
    # volume[ix, iy, iz] = f(x_ix, y_iy, z_iz)
    np.testing.assert_allclose(vol["volume"][2, 1, 3], lin[2] + 10 * lin[1] + 100 * lin[3], rtol=1e-6)
    assert vol["has_surface"] and abs(vol["spacing"][0] - 0.5) < 1e-12
    pc = torch.rand(100, 3) * torch.tensor([1.0, 2.0, 3.0])
    gg = plots.get_grid(pc, 8)
    assert gg["shortest_axis_index"] == 0 and gg["xyz"][0].shape[0] == 8


def test_checkpoint_wire_format(tmp_path):
    """training/checkpoints: the reference runner's directory layout and payload keys (idr_train.py:181-216),
    state_dict keys as the reference names them, round trip into a fresh model / optimizer / scheduler."""
    import os
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.training import checkpoints as ck
    torch.manual_seed(3)
    model = IDRNetwork(idr_conf("tiny"))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [1000, 1500], gamma=0.5)
    for p in model.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step(); sched.step()
    ck.save_checkpoints(str(tmp_path), 7, model, opt, sched)
    for sub in ("ModelParameters", "OptimizerParameters", "SchedulerParameters"):
        assert sorted(os.listdir(tmp_path / sub)) == ["7.pth", "latest.pth"]
    payload = torch.load(tmp_path / "ModelParameters" / "7.pth", weights_only=True)
    assert set(payload) == {"epoch", "model_state_dict"} and payload["epoch"] == 7
    keys = set(payload["model_state_dict"])
    assert "implicit_network.embed_model.embedder_obj.levels.0.embedding.weight" in keys
    assert "implicit_network.embed_model.embedder_obj.freq_encoding.B" in keys
    assert {"implicit_network.lin0.weight_g", "implicit_network.lin0.weight_v", "implicit_network.lin0.bias",
            "implicit_network.dencity_net.beta", "rendering_network.lin0.weight_g"} <= keys
    torch.manual_seed(4)
    model2 = IDRNetwork(idr_conf("tiny"))
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-4)
    sched2 = torch.optim.lr_scheduler.MultiStepLR(opt2, [1000, 1500], gamma=0.5)
    assert ck.load_checkpoints(str(tmp_path), model2, opt2, sched2, checkpoint=7) == 7
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert sched2.last_epoch == sched.last_epoch
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert all(torch.equal(s1[i]["exp_avg"], s2[i]["exp_avg"]) for i in s1)
    # the optimizer file has the REFERENCE's layout: one entry per hash-grid level (levels.{l}.embedding.weight)
    osd = torch.load(tmp_path / "OptimizerParameters" / "7.pth", weights_only=True)["optimizer_state_dict"]
    L_sdf = model.implicit_network.embed_model.embedder_obj.n_levels
    L_view = model.rendering_network.embed_model.embedder_obj.n_levels
    n_own = len(list(model.parameters()))
    assert len(osd["param_groups"][0]["params"]) == n_own + (L_sdf - 1) + (L_view - 1)
    emb = model.implicit_network.embed_model.embedder_obj
    k_table = [k for k, p in enumerate(model.parameters()) if p is emb.table][0]
    for l in range(L_sdf):
        assert osd["state"][k_table + l]["exp_avg"].shape == (int(emb.desc.rows[l]), 2)


def test_optimizer_state_from_a_reference_shaped_checkpoint(golden):
    """A handcrafted optimizer state dict in the reference's shape (its own parameter list: golden param_names of the
    reference run, one Adam entry per level) loads into ClipAdam / torch.optim.Adam over the fused table, and survives
    the round trip back (the cross-loading claim must hold for OptimizerParameters too)."""
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.training import checkpoints as ck
    ref_names = [str(n) for n in golden("idr_step_C1")["param_names"]]
    torch.manual_seed(5)
    model = IDRNetwork(idr_conf("C1"))
    layout = ck._reference_layout(model)
    own_names = [n for n, _ in model.named_parameters()]
    assert len(layout) == len(ref_names) == 55
    # same order as the reference's model.parameters(): level entries sit where the fused table sits
    for (k, rows), rn in zip(layout, ref_names):
        if rows is None:
            assert own_names[k] == rn
        else:
            assert own_names[k].endswith("embedder_obj.table") and "embedder_obj.levels." in rn
    ref_shapes = dict(zip(ref_names, [None] * len(ref_names)))
    sd_model = model.state_dict()
    rs = np.random.RandomState(0)
    state = {}
    for j, rn in enumerate(ref_names):
        if rn.endswith("dencity_net.beta"):
            continue                                          # never receives a gradient: no Adam state (reference too)
        shape = tuple(sd_model[rn].shape)
        state[j] = {"step": torch.tensor(12.0), "exp_avg": torch.from_numpy(rs.standard_normal(shape).astype(np.float32)),
                    "exp_avg_sq": torch.from_numpy(rs.uniform(0, 1, shape).astype(np.float32))}
    ref_sd = {"state": state, "param_groups": [dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                                                   maximize=False, foreach=None, capturable=False, differentiable=False,
                                                   fused=None, decoupled_weight_decay=False,
                                                   params=list(range(len(ref_names))))]}
    own_sd = ck.optimizer_state_from_reference(model, ref_sd)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    opt.load_state_dict(own_sd)
    emb = model.implicit_network.embed_model.embedder_obj
    st = opt.state[emb.table]
    off = [int(v) for v in emb.desc.row_off]
    j0 = ref_names.index("implicit_network.embed_model.embedder_obj.levels.0.embedding.weight")
    for l in range(emb.n_levels):
        assert torch.equal(st["exp_avg"][off[l]:off[l + 1]], state[j0 + l]["exp_avg"])
        assert torch.equal(st["exp_avg_sq"][off[l]:off[l + 1]], state[j0 + l]["exp_avg_sq"])
    assert float(st["step"]) == 12.0
    back = ck.optimizer_state_to_reference(model, opt.state_dict())
    assert set(back["state"]) == set(state)
    for j in state:
        assert torch.equal(back["state"][j]["exp_avg"], state[j]["exp_avg"]), ref_names[j]
    # levels that disagree on the step count cannot be fused
    bad = {"state": {j: dict(v) for j, v in state.items()}, "param_groups": ref_sd["param_groups"]}
    bad["state"][j0 + 1]["step"] = torch.tensor(13.0)
    with pytest.raises(ValueError):
        ck.optimizer_state_from_reference(model, bad)


def test_saved_optimizer_file_is_continued_by_torch_adam(tmp_path):
    """the OptimizerParameters file written from a ClipAdam run must be usable by the REFERENCE's
    torch.optim.Adam (idr_train.py:128,151-156): load_state_dict() into an Adam built over a reference-shaped
    parameter list (one parameter per hash-grid level), then step() - which reads group['weight_decay'],
    ['amsgrad'], ['maximize'] ... and raised KeyError before ClipAdam's group carried them."""
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.training import checkpoints as ck
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    torch.manual_seed(6)
    model = IDRNetwork(idr_conf("tiny"))
    opt = ClipAdam(model.parameters(), lr=1e-4, max_norm=1.0)
    for p in model.parameters():                      # Adam state as ClipAdam.step() lays it out (the step itself needs a GPU)
        if p.requires_grad and p.numel() > 1:
            opt.state[p] = {"step": None, "exp_avg": torch.full_like(p, 1e-3), "exp_avg_sq": torch.full_like(p, 1e-6)}
    ck.save_checkpoints(str(tmp_path), 3, model, opt)
    osd = torch.load(tmp_path / "OptimizerParameters" / "latest.pth", weights_only=True)["optimizer_state_dict"]
    grp = osd["param_groups"][0]
    ref_grp = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4).state_dict()["param_groups"][0]
    assert set(ref_grp) <= set(grp), sorted(set(ref_grp) - set(grp))
    # the reference's parameter list: the model's own, with every fused table replaced by its per-level tensors
    layout = ck._reference_layout(model)
    own = list(model.parameters())
    ref_params = []
    for k, rows in layout:
        if rows is None:
            ref_params.append(torch.nn.Parameter(own[k].detach().clone()))
        else:
            ref_params.append(torch.nn.Parameter(own[k].detach()[rows[0]:rows[1]].clone()))
    ref_opt = torch.optim.Adam(ref_params, lr=1e-4)
    ref_opt.load_state_dict(osd)
    before = [p.detach().clone() for p in ref_params]
    for p in ref_params:
        p.grad = torch.full_like(p, 2e-3)
    ref_opt.step()                                    # KeyError('weight_decay') before the fix
    moved = [not torch.equal(a, b.detach()) for a, b in zip(before, ref_params)]
    assert all(moved)
    k_tab = [j for j, (k, rows) in enumerate(layout) if rows is not None][0]
    assert ref_opt.state[ref_params[k_tab]]["exp_avg"].shape == ref_params[k_tab].shape


def _rand_rotation(rng):
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q @ np.diag(np.sign(np.diag(r)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def test_projection_decomposition_round_trip():
    """load_K_Rt_from_P (reference rend_util.py:25-46, there via cv2.decomposeProjectionMatrix): construct
    P = s K [R | -R C] and recover intrinsics (K / K[2,2]) and pose [R^T | C]; also the quaternion round trip."""
    from hashmodnffbanks_idr_amd.utils import rend_util
    rng = np.random.default_rng(0)
    for trial in range(20):
        K = np.array([[800 + 50 * rng.random(), 0.3 * rng.standard_normal(), 300 + 40 * rng.random()],
                      [0, 790 + 50 * rng.random(), 250 + 30 * rng.random()], [0, 0, 1.0]])
        R, C = _rand_rotation(rng), rng.standard_normal(3) * 2
        P = (0.5 + rng.random()) * K @ np.concatenate([R, (-R @ C)[:, None]], 1)
        intr, pose = rend_util.load_K_Rt_from_P(None, P.astype(np.float32))
        np.testing.assert_allclose(intr[:3, :3], K, rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(pose[:3, :3], R.T, atol=2e-5)
        np.testing.assert_allclose(pose[:3, 3], C, rtol=2e-4, atol=2e-4)
        assert pose.dtype == np.float32 and intr.shape == (4, 4) and intr[3, 3] == 1
    Rb = torch.from_numpy(np.stack([_rand_rotation(rng) for _ in range(8)])).float()
    q = rend_util.rot_to_quat(Rb)
    ok = (1.0 + Rb[:, 0, 0] + Rb[:, 1, 1] + Rb[:, 2, 2]) > 0.1      # the formula divides by w
    np.testing.assert_allclose(rend_util.quat_to_rot(q)[ok].numpy(), Rb[ok].numpy(), atol=1e-5)


def test_scene_dataset(tmp_path):
    """datasets/scene_dataset.SceneDataset on a synthetic two-image scan in the DTU layout (scene_dataset.py:8-117)."""
    from PIL import Image
    from hashmodnffbanks_idr_amd.datasets.scene_dataset import SceneDataset
    H, W = 6, 8
    scan = tmp_path / "DTU" / "scan65"
    (scan / "image").mkdir(parents=True)
    (scan / "mask").mkdir()
    rng = np.random.default_rng(1)
    cams, imgs, masks = {}, [], []
    for i in range(2):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        msk = (rng.random((H, W)) > 0.5).astype(np.uint8) * 255
        Image.fromarray(img).save(scan / "image" / f"{i:06d}.png")
        Image.fromarray(np.stack([msk] * 3, -1)).save(scan / "mask" / f"{i:03d}.png")
        imgs.append(img); masks.append(msk > 127)
        K = np.array([[50.0, 0, 4], [0, 50, 3], [0, 0, 1]])
        R, C = _rand_rotation(rng), rng.standard_normal(3)
        wm = np.eye(4); wm[:3] = K @ np.concatenate([R, (-R @ C)[:, None]], 1)
        sm = np.diag([2.0, 2.0, 2.0, 1.0]); sm[:3, 3] = [0.1, -0.2, 0.3]
        cams[f"world_mat_{i}"], cams[f"scale_mat_{i}"] = wm, sm
    np.savez(scan / "cameras.npz", **cams)
    ds = SceneDataset(False, "DTU", [H, W], scan_id=65, root=str(tmp_path))
    assert len(ds) == 2 and ds.total_pixels == H * W
    idx, sample, gt = ds[1]
    assert sample["uv"].shape == (H * W, 2) and sample["intrinsics"].shape == (4, 4) and sample["pose"].shape == (4, 4)
    np.testing.assert_array_equal(sample["uv"][W + 2].numpy(), [2.0, 1.0])      # pixel (row 1, col 2) -> (x=2, y=1)
    np.testing.assert_allclose(gt["rgb"].numpy(), imgs[1].reshape(-1, 3) / 255.0 * 2 - 1, atol=1e-6)
    np.testing.assert_array_equal(sample["object_mask"].numpy(), masks[1].reshape(-1))
    ds.change_sampling_idx(10)
    _, s2, g2 = ds[0]
    assert s2["uv"].shape == (10, 2) and g2["rgb"].shape == (10, 3) and s2["object_mask"].shape == (10,)
    batch = ds.collate_fn([ds[0], ds[1]])
    assert batch[0].tolist() == [0, 1] and batch[1]["uv"].shape == (2, 10, 2) and batch[2]["rgb"].shape == (2, 10, 3)
    # the pose is the camera centre / rotation of P = world_mat @ scale_mat (scaled world)
    P = (cams["world_mat_0"] @ cams["scale_mat_0"])[:3]
    centre = -np.linalg.solve(P[:, :3], P[:, 3])
    np.testing.assert_allclose(ds.pose_all[0][:3, 3].numpy(), centre, rtol=1e-4, atol=1e-4)
    assert ds.get_gt_pose().shape == (2, 4, 4) and ds.get_scale_mat().shape == (4, 4)


def test_scene_dataset_on_the_reference_dummy_scan():
    """The reference's own scene (data/dummy/scan0): its cameras.npz unmodified plus 48x64 windows of images / masks 000
    and 001 (tests/golden/dummy_scan0; data files only).  The npz also stores each view's intrinsics (camera_mat_i)
    and the inverse projection (world_mat_inv_i), so the P = K [R | t] decomposition the dataset performs
    (scene_dataset.py:40-52, rend_util.py:25-46 - cv2.decomposeProjectionMatrix in the reference) is checked against
    the scene's own ground truth, including the scale_mat normalisation of DATA_CONVENTION.md."""
    import os
    from PIL import Image
    from hashmodnffbanks_idr_amd.datasets.scene_dataset import SceneDataset
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dummy_scan0")
    cams = np.load(os.path.join(root, "cameras.npz"))
    n_views = len([k for k in cams.files if k.startswith("world_mat_") and "inv" not in k])
    assert n_views == 11

    class AllViews(SceneDataset):      # the fixture ships two images; the camera file describes all eleven views
        def _cameras(self, cam_file, scaled):
            self.n_images, keep = n_views, self.n_images
            try:
                yield from super()._cameras(cam_file, scaled)
            finally:
                self.n_images = keep

    H, W = 48, 64
    # layout <root>/<data_dir>/scan<id>: tests/golden / "." / dummy_scan0 does not match 'scan{id}', so link it
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "dummy"))
        os.symlink(root, os.path.join(tmp, "dummy", "scan0"))
        ds = AllViews(False, "dummy", [H, W], scan_id=0, root=tmp)
        assert len(ds) == 2 and len(ds.pose_all) == n_views
        for i in range(n_views):
            K = cams[f"camera_mat_{i}"].astype(np.float64)
            S = cams[f"scale_mat_{i}"].astype(np.float64)
            Winv = cams[f"world_mat_inv_{i}"].astype(np.float64)
            intr, pose = ds.intrinsics_all[i].numpy().astype(np.float64), ds.pose_all[i].numpy().astype(np.float64)
            np.testing.assert_allclose(intr[:3, :3], K[:3, :3], rtol=2e-4, atol=5e-2)      # focal ~1875 px
            centre_world = Winv[:3, 3] / Winv[3, 3]                                        # P C = 0
            centre_scaled = np.linalg.solve(S, np.append(centre_world, 1.0))[:3]           # scaled frame = S^-1 world
            np.testing.assert_allclose(pose[:3, 3], centre_scaled, rtol=1e-3, atol=2e-3)
            R_wc = np.linalg.inv(K[:3, :3]) @ cams[f"world_mat_{i}"].astype(np.float64)[:3, :3]
            R_wc /= np.cbrt(np.linalg.det(R_wc))
            np.testing.assert_allclose(pose[:3, :3], R_wc.T, atol=2e-4)                    # camera-to-world rotation
            assert abs(np.linalg.det(pose[:3, :3]) - 1) < 1e-4
            # the cameras of a normalised scene look at the unit sphere from outside it
            assert 1.0 < np.linalg.norm(pose[:3, 3]) < 10.0
        idx, sample, gt = ds[1]
        img = np.asarray(Image.open(os.path.join(root, "image", "001.png")))
        msk = np.asarray(Image.open(os.path.join(root, "mask", "001.png")))
        np.testing.assert_allclose(gt["rgb"].numpy(), img.reshape(-1, 3) / 255.0 * 2 - 1, atol=1e-6)
        np.testing.assert_array_equal(sample["object_mask"].numpy(), (msk.reshape(-1) > 127.5))
        assert 0.3 < sample["object_mask"].float().mean() < 0.7          # the window straddles the silhouette
        ds.change_sampling_idx(256)
        _, s2, g2 = ds[0]
        assert s2["uv"].shape == (256, 2) and g2["rgb"].shape == (256, 3)
