"""Full IDRNetwork.forward + IDRLoss + backward + Adam against the values captured from the reference run:
config-1 shape (256 rays, 3 steps; tests/golden/idr_step_C1.npz) and the benchmarked config-2 shape
(L=16, T=2^19 -> E=67, 2048 rays, 1 step; idr_step_C2.npz), eager and as the captured HIP graph."""
import numpy as np
import pytest
import torch

from helpers import make_idr

pytestmark = pytest.mark.gpu

CASES = [("C1", 3), ("C2", 1), ("C4", 1)]      # C4: BASELINE configs[3] per-GPU shape (T = 2^22)


def _model(g, cfg):
    return make_idr(cfg, int(g["seed"]), float(g["bias"]) if "bias" in g.files else 0.6)


def _close_mostly(a, b, rtol, atol, max_bad=0.005, what=""):
    """The SDF is discontinuous across voxel faces (hash features are piecewise constant in the
    reference's frac mode), so a ray point that differs in the last bits may sit in the neighbouring
    voxel: allow a tiny fraction of outliers, everything else tight.  Prints the observed error."""
    err = np.abs(a - b)
    bad = err > atol + rtol * np.abs(b)
    rel = err / (np.abs(b) + atol / max(rtol, 1e-30))
    print(f"    {what}: max |d| {err.max():.3e}, max |d|/(|ref| + atol/rtol) {rel.max():.3e} (rtol {rtol:g}), "
          f"{int(bad.sum())} / {bad.size} outside")
    assert bad.mean() <= max_bad, f"{what}: {bad.sum()} / {bad.size} outside tolerance, max {err.max()}"


def _check_first_step(g, out, lo, gn, model, n_rays, tag, max_flips=2):
    """Everything the reference recorded for its first iteration.  Rays whose network_object_mask differs from the
    reference's (a threshold decision on SDF values that differ in the last bits; at most 2 allowed) are masked
    out of the per-ray comparisons and widen the tolerance of the sums they enter - the checks ALWAYS run."""
    ref_mask = g["s0:network_object_mask"]
    flip = out["network_object_mask"].cpu().numpy() != ref_mask
    mism = int(flip.sum())
    print(f"[{tag}] network_object_mask mismatches: {mism} / {ref_mask.size}")
    assert mism <= max_flips, f"{mism} network_object_mask mismatches"
    keep = ~flip
    per_ray = mism / float(n_rays)
    from helpers import pin
    pkey = "step:" + tag.replace(" ", "_")
    for k, share in (("loss", 8.0), ("rgb_loss", 4.0), ("eikonal_loss", 2.0), ("mask_loss", 8.0)):
        ref = float(g[f"s0:{k}"])
        rel = abs(lo[k].item() - ref) / (abs(ref) + 1e-12)
        print(f"    {k}: {lo[k].item():.7g} vs {ref:.7g}  rel {rel:.2e}")
        assert abs(lo[k].item() - ref) <= (2e-4 + share * per_ray) * abs(ref) + 1e-6, (k, lo[k].item(), ref)
        # (a ray on the other side of a tracing threshold moves the sums it enters: not pinned then.  Floors: the step's
        #  inputs are the tracer's hit points, and the hash features are piecewise constant in them - a last-bit change of
        #  the no-grad SDF kernel moves a few points across voxel faces and these sums by ~1e-5 relative)
        if mism == 0:
            pin(f"{pkey}:{k}_rel", rel, floor=3e-5)
    ref_gn = float(g["s0:total_grad_norm"])
    print(f"    total grad norm: {gn:.6g} vs {ref_gn:.6g}  rel {abs(gn - ref_gn) / ref_gn:.2e}")
    assert abs(gn - ref_gn) <= (2e-3 + 8.0 * per_ray) * ref_gn, (gn, ref_gn)
    if mism == 0:
        pin(f"{pkey}:total_grad_norm_rel", abs(gn - ref_gn) / ref_gn, floor=1e-4)
    n_eik = n_rays // 2
    keep_g = np.concatenate([np.ones(n_eik, bool), keep])      # grad_theta rows: eikonal samples, then the ray points
    # (1 % outliers allowed on the per-ray SDF: 0.2 - 0.54 % of the ray points of the T = 2^19 / 2^22 grids sit close
    #  enough to a voxel face of some level for a last-bit difference of the point to change its voxel)
    _close_mostly(out["sdf_output"].detach().cpu().numpy()[keep], g["s0:sdf_output"][keep], 1e-4, 2e-5, max_bad=0.01,
                  what="sdf_output")
    _close_mostly(out["grad_theta"].detach().cpu().numpy()[keep_g], g["s0:grad_theta"][keep_g], 1e-3, 2e-4,
                  what="grad_theta")
    _close_mostly(out["rgb_values"].detach().cpu().numpy()[keep], g["s0:rgb_values"][keep], 1e-3, 2e-4,
                  what="rgb_values")
    emb = model.implicit_network.embed_model.embedder_obj
    emb = getattr(emb, "grid_enc", emb)          # filter-bank embedders own a hash grid (grid_enc)
    off = emb.desc.row_off
    gtol = 5e-3 + 8.0 * per_ray
    worst = 0.0
    for name, p in model.named_parameters():
        if name.startswith("implicit_network.") and name.endswith(".table"):
            stem = name[:-len("table")]
            for l in range(emb.n_levels):
                ref = float(g[f"s0:gradnorm:{stem}levels.{l}.embedding.weight"])
                got = 0.0 if p.grad is None else p.grad[int(off[l]):int(off[l + 1])].double().norm().item()
                if ref <= 0:
                    assert got == 0.0, (name, l, got)        # a level that never reaches the output (filter banks)
                    continue
                worst = max(worst, abs(got - ref) / (ref + 1e-30))
                assert abs(got - ref) <= gtol * ref + 1e-9, (name, l, got, ref)
        elif name.endswith("embedder_obj.table"):
            continue
        else:
            ref = float(g[f"s0:gradnorm:{name}"])
            if ref < 0:
                assert p.grad is None
                continue
            got = p.grad.double().norm().item()
            worst = max(worst, abs(got - ref) / (ref + 1e-30))
            assert abs(got - ref) <= gtol * ref + 1e-9, (name, got, ref)
    print(f"    per-parameter gradient norms: worst rel {worst:.2e} (tolerance {gtol:.1e})")
    if mism == 0:
        pin(f"{pkey}:param_grad_norm_worst_rel", worst, floor=2e-4)
    return mism


@pytest.mark.parametrize("cfg,n_steps", CASES)
@pytest.mark.parametrize("merge", [True, False])
def test_idr_training_steps(golden, merge, cfg, n_steps):
    """merge=False runs the reference's exact evaluation structure (three SDF-network evaluations of the
    ray points); merge=True the single merged evaluation - both must reproduce the reference run."""
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    g = golden(f"idr_step_{cfg}")
    model = _model(g, cfg)
    model.merge_evaluations = merge
    model.train()
    inp = {k: torch.from_numpy(g[k]).cuda() for k in ("intrinsics", "uv", "pose", "object_mask")}
    gt = {"rgb": torch.from_numpy(g["rgb_gt"]).cuda()}
    n_rays = g["uv"].shape[1]
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    opt = torch.optim.Adam(model.parameters(), lr=1.0e-4)
    for step in range(n_steps):
        torch.manual_seed(1000 + step)  # same CPU RNG stream as the reference run: steps, then eikonal points
        out = model(inp)
        lo = loss_fn(out, gt)
        opt.zero_grad()
        lo["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        if step == 0:
            mism = _check_first_step(g, out, lo, gn.item(), model, n_rays, f"{cfg} eager merge={merge}")
        else:
            # Adam's first updates are lr*sign(g): a gradient entry at noise level may flip sign, so the
            # trajectories separate by O(lr) in a few weights; require loss-curve agreement instead.
            ref_mask = g[f"s{step}:network_object_mask"]
            mism_s = (out["network_object_mask"].cpu().numpy() != ref_mask).sum()
            assert mism_s <= 0.05 * ref_mask.size, f"step {step}: {mism_s} mask mismatches"
            for k in ("loss", "eikonal_loss", "mask_loss"):
                ref = float(g[f"s{step}:{k}"])
                assert abs(lo[k].item() - ref) <= 0.03 * abs(ref) + 1e-4, (step, k, lo[k].item(), ref)
        before = {n: p.detach().clone() for n, p in model.named_parameters()} if step == 0 else None
        opt.step()
        if step == 0:
            # parameters after one Adam step (lr 1e-4): sampled entries, allow a handful of sign flips
            # (more of them when a ray took the other branch: its gradient share moves entries at noise level)
            worst = 0.0
            for name, p in model.named_parameters():
                if name.endswith("embedder_obj.table"):
                    continue
                idx = g[f"s0:pidx:{name}"]
                ref = g[f"s0:pval:{name}"]
                got = p.detach().cpu().numpy().reshape(-1)[idx]
                bad = np.abs(got - ref) > 2e-6 + 1e-5 * np.abs(ref)
                worst = max(worst, float(bad.mean()))
                # Adam's first update is lr * g / (|g| + 1e-8).  Entries whose reference update is a full +-lr have
                # |g| >> eps and must agree firmly; entries whose clipped gradient is of the order of eps move by less,
                # and a relative gradient error of 1e-4 there shifts the update by a few 1e-6 (seen on lin8.bias at
                # C2, where every launch-order or rounding change moves one or two of the 64 sampled entries): those
                # are only bounded by the size of one update
                upd_ref = np.abs(ref - before[name].cpu().numpy().reshape(-1)[idx])
                firm = upd_ref >= 0.9e-4
                if firm.any():
                    assert bad[firm].mean() <= (0.05 if mism == 0 else 0.25), (name, bad[firm].sum(), firm.sum())
                # the 5 % bound holds for every FIRM entry of every tensor (above); the relaxed bound applies ONLY to the
                # entries in Adam's eps regime (reference update below 0.9 lr) - and says which tensors used it
                soft = ~firm
                if soft.any() and bad[soft].any():
                    print(f"    [relaxed bound] {name}: {int(bad[soft].sum())} / {int(soft.sum())} eps-regime entries off "
                          f"({int(bad[firm].sum())} / {int(firm.sum())} firm entries off)")
                    # (a count bound with some room for tensors that have only a handful of such entries - lin8.bias at
                    #  C4 has 9, of which 2 - 4 move with the arrival order of the weight-gradient atomics, r3ah - and,
                    #  below, every entry within one update of the reference)
                    assert bad[soft].sum() <= max((0.35 if mism == 0 else 0.5) * soft.sum(), 5), \
                        (name, bad[soft].sum(), soft.sum())
                    assert np.abs(got - ref)[soft].max() <= 1.1e-4, name
                assert np.abs(got - ref).max() <= 2.5e-4, name
            print(f"    parameters after one Adam step: worst fraction of sampled entries off by a sign flip {worst:.3f}")


@pytest.mark.parametrize("coarse", [None, "f16x2"])
def test_graphed_step_C2_matches_reference(golden, coarse):
    """ONE captured-graph iteration at the benchmarked configuration (C2, 2048 rays) against the reference's own
    first iteration on the same random draws.  lr = 0 keeps the parameters at their initial values through the two
    eager warm-up iterations and the capture, and the CPU generator is re-seeded before every step, so the first
    REPLAYED iteration computes exactly what the reference's step 0 did.
    coarse = "f16x2": the same iteration with the tracer's coarse scans on the split-operand kernel (22-bit operands on
    the 16-bit matrix cores) is held to the SAME comparison with the reference's recorded iteration - mask flips, loss
    terms, gradient norms, per-ray outputs (bench.py's split_f16x2_leg)."""
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    g = golden("idr_step_C2")
    model = _model(g, "C2")
    model.implicit_network.coarse_split = coarse
    model.train()
    inp = {k: torch.from_numpy(g[k]).cuda() for k in ("intrinsics", "uv", "pose", "object_mask")}
    gt = {"rgb": torch.from_numpy(g["rgb_gt"]).cuda()}
    n_rays = g["uv"].shape[1]
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    opt = ClipAdam(model.parameters(), lr=0.0, max_norm=1.0)
    stepper = GraphedTrainStep(model, loss_fn, opt, warmup=2)
    for it in range(4):
        torch.manual_seed(1000)
        out, lo = stepper.step(inp, gt)
    assert stepper.g_fb is not None, "graph capture fell back to eager"
    torch.cuda.synchronize()
    assert np.array_equal(stepper.static["steps"].cpu().numpy(), g["s0:draw0"])       # the reference's two draws
    assert np.array_equal(stepper.static["eik"].cpu().numpy(), g["s0:draw1"])
    st = model.ray_tracer.last_stats
    assert st["unfinished"] == 0 and st.get("nonfinite", 0) == 0
    _check_first_step(g, out, lo, float(opt.last_grad_norm.item()), model, n_rays,
                      "C2 captured graph" + (f" {coarse} coarse scans" if coarse else ""))


@pytest.mark.parametrize("tag", ["C3", "C5"])
@pytest.mark.parametrize("graphed", [False, True])
def test_idr_step_filter_bank_configs(golden, tag, graphed):
    """BASELINE configs[2] ('FFB', 4096 rays) and configs[4] ('StyleModNFFB', 2048 rays; fp32): one full iteration
    against the reference's recorded iteration - eager (dynamic shapes) and as the captured graph.  sin(30 .) in the
    trunk amplifies last-bit differences of the embedding, so a few more rays than in the hash-grid configurations may
    land on the other side of a tracing threshold (<= 0.3 % allowed)."""
    from helpers import make_idr_nffb
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    g = golden(f"idr_step_{tag}")
    model = make_idr_nffb(str(g["embed_type"]), int(g["seed"]))
    model.train()
    inp = {k: torch.from_numpy(g[k]).cuda() for k in ("intrinsics", "uv", "pose", "object_mask")}
    gt = {"rgb": torch.from_numpy(g["rgb_gt"]).cuda()}
    n_rays = g["uv"].shape[1]
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    flips = max(2, int(0.003 * n_rays))
    if graphed:
        opt = ClipAdam(model.parameters(), lr=0.0, max_norm=1.0)
        stepper = GraphedTrainStep(model, loss_fn, opt, warmup=2)
        for _ in range(4):
            torch.manual_seed(1000)
            out, lo = stepper.step(inp, gt)
        assert stepper.g_fb is not None, "graph capture fell back to eager"
        torch.cuda.synchronize()
        st = model.ray_tracer.last_stats
        assert st["unfinished"] == 0 and st["nonfinite"] == 0
        _check_first_step(g, out, lo, float(opt.last_grad_norm.item()), model, n_rays, f"{tag} captured graph", flips)
    else:
        torch.manual_seed(1000)
        out = model(inp)
        lo = loss_fn(out, gt)
        model.zero_grad()
        lo["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        _check_first_step(g, out, lo, gn.item(), model, n_rays, f"{tag} eager", flips)


def test_idr_eval_forward(golden):
    """model.eval(); model(input) - the contract evaluation/eval.py relies on (grad_theta is None)."""
    g = golden("idr_eval_C1")
    model = _model(g, "C1")
    model.eval()
    inp = {k: torch.from_numpy(g[k]).cuda() for k in ("intrinsics", "uv", "pose", "object_mask")}
    out = model(inp)
    assert out["grad_theta"] is None
    flip = out["network_object_mask"].cpu().numpy() != g["network_object_mask"]
    print(f"eval forward: {int(flip.sum())} network_object_mask mismatches")
    assert flip.sum() <= 2
    keep = ~flip
    _close_mostly(out["points"].detach().cpu().numpy()[keep], g["points"][keep], 1e-4, 2e-5, what="points")
    _close_mostly(out["sdf_output"].detach().cpu().numpy()[keep], g["sdf_output"][keep], 1e-4, 2e-5, what="sdf_output")
    _close_mostly(out["rgb_values"].detach().cpu().numpy()[keep], g["rgb_values"][keep], 1e-3, 2e-4, what="rgb_values")
