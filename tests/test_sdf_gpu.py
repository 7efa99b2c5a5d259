"""GPU parity of the SDF network: fused no-grad kernel, GEMM autograd path, gradient() and the
eikonal double backward, against fixtures generated from the reference."""
import numpy as np
import pytest
import torch

import params as P
from helpers import make_implicit
from oracle import c_oracle as O

pytestmark = pytest.mark.gpu

CASES = [("full", "C1"), ("init", "C1"), ("narrow", "tiny"), ("C2", "C2")]   # C2 = the benchmarked width (E = 67)


def _net(g, cfg):
    return make_implicit(cfg, tuple(g["hidden"].tolist()), int(g["fvs"]), int(g["seed"]), float(g["perturb"]),
                         float(g["table_scale"]))


def _close(a, b, rtol=1e-5, atol=2e-6, what=""):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


@pytest.mark.parametrize("tile", [4, 8, 16, 64])
@pytest.mark.parametrize("tag,cfg", CASES)
def test_fused_forward_golden(golden, tag, cfg, tile):
    g = golden(f"sdf_{tag}")
    net = _net(g, cfg)
    net.sdf_tile_points = tile
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        out = net(x)
        sdf = net.sdf(x)
    _close(out.cpu().numpy(), g["out"], what="fused full output")
    _close(sdf.cpu().numpy(), g["out"][:, 0], what="fused sdf-only output")


@pytest.mark.parametrize("tile", [0, 4, 8, 16, 64])
@pytest.mark.parametrize("n", [1, 15, 17, 63, 64, 65, 1000, 64 * 300 + 5])
def test_fused_forward_vs_oracle_ragged(golden, n, tile):
    g = golden("sdf_full")
    net = _net(g, "C1")
    net.sdf_tile_points = tile
    L, T, b, d = P.CONFIGS["C1"]
    seed = int(g["seed"])
    levels, B, _, _ = P.make_embedder_state(seed, "C1", float(g["table_scale"]))
    prm = P.make_sdf_params(seed + 7, 3 + 4 * L, (512,) * 8, 257, (4,), 0.6, float(g["perturb"]), 0.1)
    orc = O.SdfOracle(O.Grid(L, T, b, d), np.concatenate(levels, 0), B, prm)
    x = P.make_points(n + 1, n, -1.05, 1.05)
    with torch.no_grad():
        sdf = net.sdf(torch.from_numpy(x).cuda()).cpu().numpy()
        full = net(torch.from_numpy(x[:200]).cuda()).cpu().numpy() if n >= 200 else None
    ref = orc(x)
    _close(sdf, ref[:, 0], what="sdf-only vs oracle")
    if full is not None:
        _close(full, ref[:200], what="full vs oracle")


_C2_ORACLE = {}


def _c2_oracle(g, n):
    """C oracle at the benchmarked configuration (L=16, T=2^19 -> E=67, layer 3 = 445 wide, skip K = 445+67),
    evaluated once per batch size and shared by the tile-size cases."""
    if n not in _C2_ORACLE:
        L, T, b, d = P.CONFIGS["C2"]
        seed = int(g["seed"])
        levels, B, _, _ = P.make_embedder_state(seed, "C2", float(g["table_scale"]))
        prm = P.make_sdf_params(seed + 7, 3 + 4 * L, (512,) * 8, 257, (4,), 0.6, float(g["perturb"]), 0.1)
        orc = O.SdfOracle(O.Grid(L, T, b, d), np.concatenate(levels, 0), B, prm)
        x = P.make_points(n + 1, n, -1.05, 1.05)
        _C2_ORACLE[n] = (x, orc(x))
    return _C2_ORACLE[n]


@pytest.mark.parametrize("tile", [0, 4, 8, 16, 64])
@pytest.mark.parametrize("n", [1, 17, 65, 2049, 4097, 64 * 300 + 5])
def test_fused_forward_vs_oracle_ragged_C2(golden, n, tile):
    """Every tile size of the fused kernel at the bench configuration, value by value against the oracle
    (rtol 1e-5): different k padding of both packed images than C1 (E = 67 -> 9 octets / 5 16-blocks)."""
    g = golden("sdf_C2")
    net = _net(g, "C2")
    net.sdf_tile_points = tile
    x, ref = _c2_oracle(g, n)
    with torch.no_grad():
        sdf = net.sdf(torch.from_numpy(x).cuda()).cpu().numpy()
        full = net(torch.from_numpy(x[:300]).cuda()).cpu().numpy() if n >= 300 else None
    err = np.abs(sdf - ref[:, 0]) / (np.abs(ref[:, 0]) + 2e-6 / 1e-5)
    print(f"C2 fused sdf n={n} tile={tile}: max |d| / (|ref| + 0.2) = {err.max():.3e}")
    _close(sdf, ref[:, 0], what="sdf-only vs oracle (C2)")
    if full is not None:
        _close(full, ref[:300], what="full vs oracle (C2)")


@pytest.mark.parametrize("tag,cfg", CASES)
def test_grad_path_forward_and_first_order(golden, tag, cfg):
    g = golden(f"sdf_{tag}")
    net = _net(g, cfg)
    net.train()
    x = torch.from_numpy(g["x"].copy()).cuda().requires_grad_(True)
    out = net(x)
    _close(out.detach().cpu().numpy(), g["out"], what="grad-path forward")
    (out * torch.from_numpy(g["R"]).cuda()).sum().backward()
    # d/dx sums ~257 signed terms: tolerance relative to the gradient's scale, not per element
    _close(x.grad.cpu().numpy(), g["dx_first"], rtol=2e-5, atol=1e-5 * float(np.abs(g["dx_first"]).max()),
           what="d/dx")
    _check_param_grads(net, g, "g1", pin_name=f"sdf:{tag}:g1")


def _check_param_grads(net, g, label, pin_name=None):
    """every parameter gradient against the reference's (norms; full tensors or sampled entries); the worst observed
    errors over all parameters are pinned (helpers.pin: <= 3x the value recorded on the GPU box)"""
    from helpers import pin
    emb = net.embed_model.embedder_obj
    off = emb.desc.row_off
    worst = [0.0, 0.0]
    for name, p in net.named_parameters():
        if p.grad is None:
            continue
        arr = p.grad.detach().cpu().numpy()
        if name.endswith("embedder_obj.table"):
            for l in range(emb.n_levels):
                _cmp(arr[int(off[l]):int(off[l + 1])], g, f"{label}:table{l}", worst)
        else:
            _cmp(arr, g, f"{label}:{name}", worst)
    print(f"    {label}: worst gradient-norm rel error {worst[0]:.3e}, worst entry error / tensor scale {worst[1]:.3e}")
    if pin_name:
        pin(pin_name + ":grad_norm_rel", worst[0])
        pin(pin_name + ":grad_entry_rel_to_scale", worst[1])


def _cmp(arr, g, key, worst=None):
    ref_norm = float(g[key + ":norm"])
    got_norm = float(np.linalg.norm(arr.astype(np.float64)))
    assert abs(got_norm - ref_norm) <= 1e-4 * max(ref_norm, 1e-12) + 1e-9, (key, got_norm, ref_norm)
    scale = max(ref_norm / np.sqrt(arr.size), 1e-12)
    if key + ":full" in g.files:
        ref = g[key + ":full"]
        got = arr
    else:
        ref = g[key + ":val"]
        got = arr.reshape(-1)[g[key + ":idx"]]
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-4 * max(np.abs(ref).max(), scale), err_msg=key)
    if worst is not None:
        worst[0] = max(worst[0], abs(got_norm - ref_norm) / max(ref_norm, 1e-12))
        worst[1] = max(worst[1], float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), scale))


@pytest.mark.parametrize("tag,cfg", CASES)
def test_gradient_and_eikonal_double_backward(golden, tag, cfg):
    g = golden(f"sdf_{tag}")
    net = _net(g, cfg)
    net.train()
    x = torch.from_numpy(g["x"].copy()).cuda()
    gr = net.gradient(x)
    assert gr.shape == (x.shape[0], 1, 3)
    _close(gr.detach().cpu().numpy()[:, 0, :], g["gradient"], rtol=2e-5, atol=1e-5, what="gradient()")
    eik = ((gr[:, 0, :].norm(2, dim=1) - 1) ** 2).mean()
    assert abs(eik.item() - float(g["eik"])) <= 1e-5 * abs(float(g["eik"])) + 1e-7
    from helpers import pin
    pin(f"sdf:{tag}:gradient_rel_to_scale",
        float(np.abs(gr.detach().cpu().numpy()[:, 0, :] - g["gradient"]).max() / np.abs(g["gradient"]).max()))
    pin(f"sdf:{tag}:eikonal_rel", abs(eik.item() - float(g["eik"])) / abs(float(g["eik"])))
    eik.backward()
    _check_param_grads(net, g, "g2", pin_name=f"sdf:{tag}:g2")


def test_gemm_shapes_vs_torch():
    from hashmodnffbanks_idr_amd import ops
    rs = np.random.RandomState(0)
    for (M, N, K) in [(1, 1, 1), (100, 445, 67), (300, 257, 512), (64, 64, 16), (513, 130, 33), (2048, 512, 512),
                      (3072, 445, 512), (2700, 512, 512), (130, 67, 256), (445, 512, 1024), (65, 1, 128),
                      (3072, 512, 445), (2048, 512, 257), (777, 300, 190)]:
        a = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).cuda()
        b = torch.from_numpy(rs.standard_normal((K, N)).astype(np.float32)).cuda()
        bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).cuda()
        ref = (a.double() @ b.double() + bias.double()).float()
        tol = dict(rtol=1e-5, atol=2e-5 * np.sqrt(K))
        np.testing.assert_allclose(ops.gemm(a, b, bias).cpu(), ref.cpu(), **tol)
        np.testing.assert_allclose(ops.gemm(a.t().contiguous(), b, bias, trans_a=True).cpu(), ref.cpu(), **tol)
        np.testing.assert_allclose(ops.gemm(a, b.t().contiguous(), bias, trans_b=True).cpu(), ref.cpu(), **tol)
        np.testing.assert_allclose(ops.gemm(a.t().contiguous(), b.t().contiguous(), bias, True, True).cpu(),
                                   ref.cpu(), **tol)
    # split-K shape of the weight gradient: small M x N, long K
    a = torch.from_numpy(rs.standard_normal((6000, 512)).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.standard_normal((6000, 445)).astype(np.float32)).cuda()
    ref = (a.double().t() @ b.double()).float()
    np.testing.assert_allclose(ops.gemm(a, b, None, trans_a=True).cpu(), ref.cpu(), rtol=1e-5, atol=2e-3)


def test_matmul_double_backward_vs_torch():
    from hashmodnffbanks_idr_amd import ops
    torch.manual_seed(0)
    x = torch.randn(37, 19, device="cuda", requires_grad=True)
    w = torch.randn(23, 19, device="cuda", requires_grad=True)
    b = torch.randn(23, device="cuda", requires_grad=True)

    def run(lin):
        y = torch.tanh(lin(x, w, b))
        (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
        loss = (gx ** 2).sum() + y.pow(2).sum()
        return torch.autograd.grad(loss, [x, w, b])

    got = run(ops.linear)
    ref = run(torch.nn.functional.linear)
    for a_, b_ in zip(got, ref):
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_fused_device_side_count():
    """n_dev: the kernel evaluates min(n, *n_dev) points without a host round trip."""
    from hashmodnffbanks_idr_amd import ops
    net = make_implicit("tiny", (64,) * 8, 16, 3, 0.5, 0.5)
    emb = net.embed_model.embedder_obj
    x = torch.from_numpy(P.make_points(2, 500)).cuda()
    full = net.sdf(x)
    for tile in (4, 8, 16, 64):
        n_dev = torch.tensor([137], dtype=torch.int32, device="cuda")
        pk = net.packed_weights()
        res = ops.sdf_fwd(emb.desc, pk, x, emb.table.detach(), emb.freq_encoding.B, 0, sdf_only=True,
                          tile_points=tile, n_dev=n_dev)
        assert torch.allclose(res[:137], full[:137], rtol=1e-6, atol=1e-7)


def test_fused_softplus_matches_torch_through_second_order():
    """Reference = torch's own softplus in float64 (truth) and in float32.  torch's fp32 double backward
    evaluates beta*s*(1-s) with cancellation (up to ~8e-5 rel error vs fp64); the fused kernel uses
    beta*e/(e+1)^2, so it must sit within 5e-6 of the fp64 truth and within 2e-4 of torch's fp32."""
    from hashmodnffbanks_idr_amd import ops
    torch.manual_seed(1)
    z0 = (torch.randn(700, 129, device="cuda") * 0.2)
    z0[0, :5] = torch.tensor([0.3, 0.2000001, 0.1999, -5.0, 0.0], device="cuda")   # around the threshold 100*z > 20

    def run(fn, dt):
        z = z0.clone().to(dt).requires_grad_(True)
        w = torch.linspace(-1, 1, 129, device="cuda", dtype=dt)
        y = fn(z)
        (g,) = torch.autograd.grad((y * w).sum(), z, create_graph=True)
        loss = (g ** 2).sum() + (y ** 2).sum()
        (gz,) = torch.autograd.grad(loss, z)
        return [t.detach().double().cpu().numpy() for t in (y, g, gz)]

    mine = run(lambda t: ops.softplus(t, 100.0, 20.0), torch.float32)
    t32 = run(lambda t: torch.nn.functional.softplus(t, beta=100, threshold=20), torch.float32)
    t64 = run(lambda t: torch.nn.functional.softplus(t, beta=100, threshold=20), torch.float64)
    for u, v, w in zip(mine, t64, t32):
        np.testing.assert_allclose(u, v, rtol=5e-6, atol=2e-6)
        np.testing.assert_allclose(u, w, rtol=2e-4, atol=2e-6)


def test_colsum_matches_torch():
    from hashmodnffbanks_idr_amd import ops
    x = torch.randn(3001, 445, device="cuda", requires_grad=True)
    s = ops.colsum(x)
    np.testing.assert_allclose(s.detach().cpu().numpy(), x.detach().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)
    (s * torch.arange(445, device="cuda")).sum().backward()
    assert torch.equal(x.grad[5], torch.arange(445, device="cuda").float())


def test_pack_kernel_matches_layout_contract():
    """hm_pack_mlp_layer vs the torch expression of the layout contract (ops.pack_mlp_layer)."""
    from hashmodnffbanks_idr_amd import ops
    torch.manual_seed(0)
    E = 67
    Ws = [torch.randn(512, E), torch.randn(445, 512), torch.randn(512, 445 + E), torch.randn(257, 512)]
    bs = [torch.randn(w.shape[0]) for w in Ws]
    pk = ops.PackedSdf([w.cuda() for w in Ws], [b.cuda() for b in bs], E, (2,), 0.9001)
    segs = [[(1, E)], [(0, 512)], [(0, 445), (1, E)], [(0, 512)]]
    for l, W in enumerate(Ws):
        ref8, n_tiles, octs, srcs = ops.pack_mlp_layer(W, segs[l])
        ref16, _, blk, _ = ops.pack_mlp_layer(W, segs[l], kblock=16)
        img8, img16, bpad = pk.bufs[l]
        assert torch.equal(img8.cpu(), ref8.reshape(-1)), l
        assert torch.equal(img16.cpu(), ref16.reshape(-1)), l
        assert torch.equal(bpad.cpu()[:W.shape[0]], bs[l]) and float(bpad.cpu()[W.shape[0]:].abs().sum()) == 0.0
        ly = pk.desc.layer[l]
        assert (ly.n_tiles, list(ly.seg_octets), list(ly.seg_blocks16), list(ly.seg_src)[:len(segs[l])]) == \
            (n_tiles, octs, blk, srcs[:len(segs[l])])


@pytest.mark.parametrize("tag,cfg", [("full", "C1"), ("narrow", "tiny")])
def test_fused_mlp_grad_node_matches_generic_autograd(golden, tag, cfg):
    """mlp_grad.sdf_mlp (analytic reverse-over-reverse) vs the generic create_graph route: outputs, input
    gradient, and every parameter / input gradient of a loss that uses both."""
    g = golden(f"sdf_{tag}")
    res = []
    for fused in (True, False):
        net = _net(g, cfg)
        net.train()
        net.use_fused_mlp_grad = fused
        x = torch.from_numpy(g["x"].copy()).cuda()
        out, gr = net.forward_with_gradient(x)
        R = torch.from_numpy(g["R"]).cuda()
        loss = ((gr[:, 0, :].norm(2, dim=1) - 1) ** 2).mean() + 0.01 * (out * R).sum() + (gr[:, 0, :] * x).sum()
        loss.backward()
        res.append((out.detach(), gr.detach(), x.grad.clone(),
                    {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    (o1, g1, x1, p1), (o2, g2, x2, p2) = res
    def close(a, b, what):
        scale = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= 2e-5 * scale, (what, (a - b).abs().max().item(), scale)
    close(o1, o2, "out"); close(g1, g2, "gradient"); close(x1, x2, "x.grad")
    assert p1.keys() == p2.keys()
    for n in p1:
        close(p1[n], p2[n], n)


def test_sdf_volume_matches_chunked_forward(golden):
    """utils/plots.sdf_volume (mesh-extraction caller of the fused kernel, reference plots.py:110-128) against
    the reference's own procedure: 10 000-point chunks of the layer-by-layer forward."""
    from hashmodnffbanks_idr_amd.utils import plots
    g = golden("sdf_full")
    net = _net(g, "C1")
    grid = plots.get_grid_uniform(33, "cuda")
    vol = plots.sdf_volume(net.sdf, grid)
    # grad mode on: ImplicitNetwork.forward takes the layer-by-layer (GEMM) route
    z = torch.cat([net(p)[:, 0].detach() for p in torch.split(grid["grid_points"], 10000, dim=0)]).cpu().numpy()
    ref = z.reshape(33, 33, 33).transpose([1, 0, 2])
    _close(vol["volume"], ref, what="sdf volume")
    assert vol["volume"].shape == (33, 33, 33)


@pytest.mark.parametrize("tile", [4, 16, 64])
def test_fused_forward_full_size_properties(golden, tile):
    """bench-size launch (2^18 points) of the fused SDF kernel: a 512-point sample against the oracle, and
    permutation equivariance bit for bit (a point's value may not depend on its tile neighbours or position)."""
    g = golden("sdf_full")
    net = _net(g, "C1")
    net.sdf_tile_points = tile
    n = 1 << 18 if tile != 4 else 1 << 16
    gen = torch.Generator(device="cpu").manual_seed(99)
    x = (torch.rand((n, 3), generator=gen) * 2.1 - 1.05).cuda()
    with torch.no_grad():
        s = net.sdf(x)
        perm = torch.randperm(n, generator=gen).cuda()
        sp = net.sdf(x[perm].contiguous())
    assert torch.equal(sp, s[perm])
    L, T, b, d = P.CONFIGS["C1"]
    seed = int(g["seed"])
    levels, B, _, _ = P.make_embedder_state(seed, "C1", float(g["table_scale"]))
    prm = P.make_sdf_params(seed + 7, 3 + 4 * L, (512,) * 8, 257, (4,), 0.6, float(g["perturb"]), 0.1)
    orc = O.SdfOracle(O.Grid(L, T, b, d), np.concatenate(levels, 0), B, prm)
    idx = torch.randint(0, n, (512,), generator=gen)
    ref = orc(x[idx.cuda()].cpu().numpy())
    _close(s[idx.cuda()].cpu().numpy(), ref[:, 0], what="sample vs oracle")


@pytest.mark.parametrize("n", [37, 64 * 256 * 2 + 5000, 64 * 256 + 8192 + 64 * 3 + 1, 204800])
def test_fused_forward_half_tile_schedule(golden, n):
    """64-point kernel, tile schedule with 32-point half tiles for the remainder round (hm_sdf.hip: `half`): a point's
    value must not depend on whether it sits in a full or a half tile - permutation equivariance bit for bit across the
    two kinds (n = 2 full rounds + 157 half tiles; a remainder above 32 points per workgroup -> full tiles; the bench's
    204 800 points = 12 rounds + 256 half tiles; 37 points = one full-width ragged tile), and the values against the
    16-point kernel's (another MFMA shape: tolerance, not bits)."""
    g = golden("sdf_C2")
    net = _net(g, "C2")
    gen = torch.Generator(device="cpu").manual_seed(5 + n)
    x = (torch.rand((n, 3), generator=gen) * 2.1 - 1.05).cuda()
    perm = torch.randperm(n, generator=gen).cuda()
    with torch.no_grad():
        net.sdf_tile_points = 64
        s = net.sdf(x)
        sp = net.sdf(x[perm].contiguous())
        full = net(x[:4096].contiguous()) if n >= 4096 else None
        net.sdf_tile_points = 16
        s16 = net.sdf(x[: min(n, 20000)].contiguous())
    assert torch.equal(sp, s[perm])
    _close(s[: s16.shape[0]].cpu().numpy(), s16.cpu().numpy(), what="64-point schedule vs 16-point tiles")
    if full is not None:
        # all output columns (last layer on the MFMA instead of the sdf-only VALU dot): same schedule, 128 half tiles
        _close(full[:, 0].cpu().numpy(), s[:4096].reshape(-1).cpu().numpy(), what="full output vs sdf-only")


def test_relu_mlp_node_matches_generic_ops():
    """mlp_grad.relu_mlp (rendering network's Linear / ReLU stack as one autograd node with ReLU GEMM epilogues)
    against the same stack on torch ops: output and first-order gradients w.r.t. input, weights and biases."""
    from hashmodnffbanks_idr_amd import mlp_grad
    g = torch.Generator(device="cpu").manual_seed(11)
    dims = [281, 512, 512, 512, 512, 3]
    for n in (2048, 77):
        Ws = [(torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i])).cuda().requires_grad_(True)
              for i in range(5)]
        bs = [(torch.randn(dims[i + 1], generator=g) * 0.1).cuda().requires_grad_(True) for i in range(5)]
        x = torch.randn(n, 281, generator=g).cuda().requires_grad_(True)
        w_out = torch.randn(n, 3, generator=g).cuda()

        def ref(xx):
            h = xx
            for l in range(5):
                h = torch.nn.functional.linear(h.double(), Ws[l].double(), bs[l].double())
                if l < 4:
                    h = torch.relu(h)
            return h

        y = mlp_grad.relu_mlp(x, Ws, bs)
        gy = torch.autograd.grad((torch.tanh(y) * w_out).sum(), [x] + Ws + bs)
        yr = ref(x)
        gr = torch.autograd.grad((torch.tanh(yr) * w_out.double()).sum(), [x] + Ws + bs)
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
        for a, b in zip(gy, gr):
            scale = float(b.abs().max()) + 1e-12
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5 * scale)
