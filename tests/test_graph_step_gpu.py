"""Static-shape forward / graph-captured training step against the dynamic (reference-structured) step."""
import copy

import numpy as np
import pytest
import torch

import bench
from helpers import idr_conf

pytestmark = pytest.mark.gpu


def _setup(seed=3):
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    torch.manual_seed(seed)
    model = IDRNetwork(idr_conf("C1")).cuda()
    with torch.no_grad():  # let the hash features matter
        model.implicit_network.lin0.weight_v[:, 3:].normal_(0, 0.02)
        model.implicit_network.embed_model.embedder_obj.table.uniform_(-0.05, 0.05)
    model.train()
    inp, gt = bench.synthetic_batch(21, 512, "cuda")
    rs = np.random.RandomState(4)
    inp["object_mask"] = torch.from_numpy(rs.uniform(0, 1, (1, 512)) < 0.8).cuda()
    gt["rgb"] = torch.from_numpy(rs.uniform(-1, 1, (1, 512, 3)).astype(np.float32)).cuda()
    return model, IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0), inp, gt


def test_static_forward_matches_dynamic_forward():
    from hashmodnffbanks_idr_amd.model.loss import idr_loss_terms as idr_loss_static
    model, loss_fn, inp, gt = _setup()
    torch.manual_seed(5)
    out_d = model(inp)
    lo_d = loss_fn(out_d, gt)
    model.zero_grad()
    lo_d["loss"].backward()
    gd = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    torch.manual_seed(5)
    steps = torch.empty(100).uniform_(0.0, 1.0).cuda()
    eik = torch.empty(256, 3).uniform_(-1.0, 1.0).cuda()
    out_s = model.forward_static(inp, eik, steps)
    lo_s = idr_loss_static(out_s, gt["rgb"], 0.1, 100.0, 50.0)
    model.zero_grad()
    lo_s["loss"].backward()
    assert torch.equal(out_s["network_object_mask"], out_d["network_object_mask"])
    for k in ("loss", "rgb_loss", "eikonal_loss", "mask_loss"):
        assert abs(lo_s[k].item() - lo_d[k].item()) <= 1e-5 * abs(lo_d[k].item()) + 1e-7, k
    np.testing.assert_allclose(out_s["rgb_values"].detach().cpu(), out_d["rgb_values"].detach().cpu(), rtol=1e-4, atol=1e-5)
    for n, p in model.named_parameters():
        if n in gd:
            a, b = p.grad, gd[n]
            scale = b.abs().max().item() + 1e-12
            assert (a - b).abs().max().item() <= 2e-4 * scale, (n, (a - b).abs().max().item(), scale)


def test_graphed_step_matches_eager_static_step():
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    model_a, loss_fn, inp, gt = _setup()
    model_b = _setup()[0]  # (nn.utils.weight_norm modules cannot be deep-copied)
    model_b.load_state_dict(model_a.state_dict())
    runs = []
    for model, use_graph in ((model_a, True), (model_b, False)):
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
        stepper = GraphedTrainStep(model, loss_fn, opt, warmup=2, use_graph=use_graph)
        torch.manual_seed(9)
        losses = []
        for _ in range(6):
            out, lo = stepper.step(inp, gt)
            losses.append(lo["loss"].item())
        runs.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()},
                     model.ray_tracer.last_stats))
    (la, pa, sa), (lb, pb, sb) = runs
    assert sa["unfinished"] == 0 and sb["unfinished"] == 0
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-3 * abs(y) + 1e-6, (la, lb)
    for n in pa:
        assert (pa[n] - pb[n]).abs().max().item() <= 5e-4, n


def test_graph_contains_no_memset_or_memcpy_nodes_of_ours(tmp_path):
    """Regression guard for the round-1 replay fault: hipMemsetAsync / hipMemcpyAsync calls recorded into the
    captured iteration (tracer cursor array, column-sum outputs, tracer statistics) lost their order under
    back-to-back replays.  The library's own fills / copies are kernels now: the captured forward+backward graph
    must hold no MEMSET node at all (torch's zero_/zeros are fill kernels) and stay a pure chain."""
    import os
    import re
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    model, loss_fn, inp, gt = _setup()
    os.environ["HM_GRAPH_DUMP"] = str(tmp_path)
    try:
        stepper = GraphedTrainStep(model, loss_fn, ClipAdam(model.parameters(), lr=1e-4), warmup=2)
        torch.manual_seed(9)
        for _ in range(4):
            out, lo = stepper.step(inp, gt)
    finally:
        os.environ.pop("HM_GRAPH_DUMP", None)
    assert stepper.g_fb is not None, "graph capture fell back to eager"
    dot = open(tmp_path / "g_fb.dot").read()
    kinds = re.findall(r'label="\{\s*\n?(\w+)\n', dot)
    assert kinds.count("KERNEL") > 100
    assert kinds.count("MEMSET") == 0, "a MEMSET node is back in the captured iteration"
    # every node of the chain with what surrounds it: names the op that recorded a copy node, should one come back
    nodes = re.findall(r'"(graph_\d+_node_\d+)"\[[^\]]*?label="\{\s*\n?(\w+)\n([^"]*)"', dot)
    def short(body):
        m = re.search(r"\| (_Z\w+)", body)
        return m.group(1)[:260] if m else re.sub(r"\s+", " ", body)[:400]
    copies = [i for i, n in enumerate(nodes) if n[1] != "KERNEL"]
    for i in copies:
        print(f"non-kernel node {nodes[i][1]}: {short(nodes[i][2])}")
        for d in (-4, -3, -2, -1, 1, 2, 3, 4):
            if 0 <= i + d < len(nodes):
                print(f"    [{d:+d}] {short(nodes[i + d][2])}")
    assert kinds.count("MEMCPY") == 0, f"{kinds.count('MEMCPY')} MEMCPY nodes in the captured iteration"
    edges = re.findall(r'"(graph_\d+_node_\d+)"\s*->\s*"(graph_\d+_node_\d+)"', dot)
    assert len(edges) == len(kinds) - 1, "captured iteration is no longer a single chain"
    opt_dot = open(tmp_path / "g_opt.dot").read()
    assert len(re.findall(r'label="\{\s*\n?KERNEL\n', opt_dot)) <= 8      # clip + Adam: begin, norm, update (+ few)
    assert np.isfinite(lo["loss"].item())


def test_run_ahead_replays_with_a_device_sync_in_between():
    """The access pattern that used to fault: replays that run ahead of the host with one device-wide
    synchronisation somewhere in between (sync_each_step=False).  Counters and loss must stay sane."""
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    model, loss_fn, inp, gt = _setup()
    stepper = GraphedTrainStep(model, loss_fn, ClipAdam(model.parameters(), lr=1e-4), warmup=2, sync_each_step=False)
    torch.manual_seed(9)
    for i in range(30):
        if i == 5:
            torch.cuda.synchronize()
        out, lo = stepper.step(inp, gt)
    torch.cuda.synchronize()
    st = model.ray_tracer.last_stats
    assert np.isfinite(lo["loss"].item())
    assert st["rays"] == 512 and st["unfinished"] == 0 and 0 <= st["sampler_rays"] <= 512
    assert 512 <= st["sdf_evals"] <= 512 * 2 * 41 + 512 * 208


def _sdf_matches_layerwise(net, x):
    """fused no-grad kernel (packed weight images) vs the layer-wise GEMM path (live parameters)"""
    with torch.no_grad():
        fused = net.sdf(x)
    with torch.enable_grad():
        layer = net(x)[:, 0].detach()      # parameters require grad -> the GEMM route, never the packed images
    err = (fused - layer).abs().max().item()
    assert err <= 1e-5, f"fused SDF kernel uses stale packed weights: max |d| = {err:.3e}"


def test_packed_weights_follow_clipadam_updates_eager():
    """ClipAdam writes the parameters through raw pointers (tensor._version does not move): the packed images of
    the fused SDF kernel must still be rebuilt after every step (cache keyed on _version only)."""
    from hashmodnffbanks_idr_amd import parallel
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    model, loss_fn, inp, gt = _setup()
    net = model.implicit_network
    x = (torch.rand(777, 3, device="cuda") * 2 - 1)
    _sdf_matches_layerwise(net, x)
    opt = ClipAdam(model.parameters(), lr=2e-3, max_norm=1.0)      # large steps: stale images would be far off
    torch.manual_seed(9)
    for _ in range(3):
        parallel.train_step(model, loss_fn, opt, inp, gt)
        _sdf_matches_layerwise(net, x)


def test_packed_weights_follow_graph_replays():
    """After graph replays (which never run Python) an eager net.sdf() must see the LAST optimizer update."""
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    model, loss_fn, inp, gt = _setup()
    net = model.implicit_network
    x = (torch.rand(777, 3, device="cuda") * 2 - 1)
    stepper = GraphedTrainStep(model, loss_fn, ClipAdam(model.parameters(), lr=2e-3, max_norm=1.0), warmup=2)
    torch.manual_seed(9)
    for i in range(6):
        stepper.step(inp, gt)
        if i >= 1:
            _sdf_matches_layerwise(net, x)
    assert stepper.g_fb is not None


def test_static_step_reusing_the_ray_rows_equals_re_evaluating_them():
    """IDRNetwork.forward_static evaluates the ray points twice (detached among the eikonal samples, then through
    SampleNetwork's re-parametrisation); with reuse_ray_rows the second evaluation takes the first one's MLP forward
    results (mlp_grad._SdfMlpRows).  The reused rows are bit-identical to the SAME rows of the big batch
    (scripts/reuse_debug.py: 0.0), while a separate 512-row evaluation differs from them by an ulp (another GEMM tile
    split: 1.2e-7 on out, 6e-8 on the gradient) - so outputs are compared to 2e-6 and gradients to 2e-5 of the
    tensor's scale (order of the fp32 atomics of the weight-gradient accumulation included)."""
    import bench
    from helpers import idr_conf
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.model.loss import idr_loss_terms
    res = []
    for reuse in (False, True):
        torch.manual_seed(0)
        model = IDRNetwork(idr_conf("C1")).cuda()
        with torch.no_grad():
            model.implicit_network.lin0.weight_v[:, 3:].normal_(0, 0.02)
            model.implicit_network.embed_model.embedder_obj.table.uniform_(-0.05, 0.05)
        model.train()
        model.reuse_ray_rows = reuse
        inp, gt = bench.synthetic_batch(7, 512, "cuda")
        rs = np.random.RandomState(3)
        inp["object_mask"] = torch.from_numpy(rs.uniform(0, 1, (1, 512)) < 0.8).cuda()
        gt["rgb"] = torch.from_numpy(rs.uniform(-1, 1, (1, 512, 3)).astype(np.float32)).cuda()
        g = torch.Generator().manual_seed(5)
        steps = torch.empty(100).uniform_(0, 1, generator=g).cuda()
        eik = torch.empty(256, 3).uniform_(-1, 1, generator=g).cuda()
        out = model.forward_static(inp, eik, steps)
        lo = idr_loss_terms(out, gt["rgb"], 0.1, 100.0, 50.0)
        lo["loss"].backward()
        res.append(({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}, lo["loss"].item(),
                    {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    (o0, l0, g0), (o1, l1, g1) = res
    for k in o0:
        if o0[k].dtype.is_floating_point:
            d = (o0[k] - o1[k]).abs().max().item()
            assert d <= 2e-6 * max(1.0, o0[k].abs().max().item()), (k, d)
        else:
            assert torch.equal(o0[k], o1[k]), k
    assert abs(l0 - l1) <= 2e-6 * abs(l0)
    assert set(g0) == set(g1)
    worst = 0.0
    for n in g0:
        scale = g0[n].abs().max().item() + 1e-30
        worst = max(worst, (g0[n] - g1[n]).abs().max().item() / scale)
    print(f"reuse of the ray rows: worst gradient difference / tensor scale {worst:.3e}")
    assert worst <= 2e-5


def test_local_table_grad_equals_autograd_accumulation(monkeypatch):
    """training/graph_step.LocalTableGrad (every table backward of the static step scatters into ONE dense gradient bound
    as table.grad) against autograd's per-node dense gradients + adds (HM_LOCAL_TABLE_GRAD=0): the same gradients after
    replayed steps (weights held: lr = 0, so both runs see the same iteration), up to the arrival order of the fp32
    atomics."""
    import bench
    from helpers import idr_conf
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    res = []
    for local in ("1", "0"):
        monkeypatch.setenv("HM_LOCAL_TABLE_GRAD", local)
        torch.manual_seed(0)
        model = IDRNetwork(idr_conf("C1")).cuda()
        with torch.no_grad():     # (the geometric initialisation zeroes the weights of the hash-feature columns)
            model.implicit_network.lin0.weight_v[:, 3:].normal_(0, 0.02)
            model.implicit_network.embed_model.embedder_obj.table.uniform_(-0.05, 0.05)
        model.train()
        opt = ClipAdam(model.parameters(), lr=0.0, max_norm=1.0)
        stepper = GraphedTrainStep(model, IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0), opt, None, warmup=2)
        assert bool(stepper.local_tables) == (local == "1")
        inp, gt = bench.synthetic_batch(7, 512, "cuda")
        torch.manual_seed(5)
        for _ in range(5):
            stepper.step(inp, gt)
        torch.cuda.synchronize()
        assert stepper.g_fb is not None
        res.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert set(res[0]) == set(res[1])
    worst = 0.0
    for n in res[0]:
        d = (res[0][n] - res[1][n]).abs().max().item()
        worst = max(worst, d / (res[1][n].abs().max().item() + 1e-30))
        if n.endswith("table"):
            rows = [int((r[n].abs().sum(1) > 0).sum()) for r in res]
            print(f"{n}: rows with a gradient {rows}")
            assert rows[0] == rows[1] and (rows[0] > 100 or not n.startswith("implicit_network"))   # (not two empty tensors)
    print(f"LocalTableGrad vs autograd accumulation, replayed step: worst gradient difference / tensor scale {worst:.3e}")
    assert worst <= 2e-5
