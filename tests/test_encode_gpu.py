"""GPU parity of the hash-grid encoder (HIP, through the C ABI) against the golden fixtures
generated from the reference and against the CPU oracle on larger seeded inputs."""
import numpy as np
import pytest
import torch

import params as P
from oracle import c_oracle as O

pytestmark = pytest.mark.gpu


def _embedder(cfg, seed, scale, frac_mode="reference"):
    from hashmodnffbanks_idr_amd.model.embeddings.hashGridEmbedding import MultiResHashGridMLP
    L, T, b, d = P.CONFIGS[cfg]
    emb = MultiResHashGridMLP(True, 3, L, 2, T, b, d, frac_mode=frac_mode).cuda()
    levels, B, res, rows = P.make_embedder_state(seed, cfg, scale)
    sd = {f"levels.{l}.embedding.weight": torch.from_numpy(np.ascontiguousarray(t)) for l, t in enumerate(levels)}
    sd["freq_encoding.B"] = torch.from_numpy(B)
    emb.load_state_dict(sd)
    return emb, np.concatenate(levels, 0), B


def test_corner_ids_bit_exact(golden):
    from hashmodnffbanks_idr_amd import ops
    g = golden("hash_ids")
    for i, (res, rows) in enumerate(g["combos"]):
        desc = ops.GridDesc([int(res)], [int(rows)], 2)
        xi, ids = ops.corner_ids(desc, 0, torch.from_numpy(g[f"x_{i}"]).cuda())
        assert np.array_equal(xi.cpu().numpy(), g[f"xi_{i}"]), (res, rows)
        assert np.array_equal(ids.cpu().numpy().astype(np.uint32), g[f"ids_{i}"]), (res, rows)


@pytest.mark.parametrize("cfg", ["C1", "C2", "shipped", "viewdir", "tiny"])
def test_encode_fwd_golden(golden, cfg):
    g = golden(f"encode_{cfg}")
    emb, _, _ = _embedder(cfg, int(g["seed"]), float(g["table_scale"]))
    with torch.no_grad():
        out = emb(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    ref = g["out"]
    L = P.CONFIGS[cfg][0]
    nf = 3 + 2 * L
    assert out.shape == ref.shape
    assert np.array_equal(out[:, :3], ref[:, :3])
    assert np.array_equal(out[:, nf:], ref[:, nf:]), "hash features must be bit-exact in reference mode"
    # Fourier part: tolerance 1e-5 rel fp32 (north star); sin/cos are O(1) so absolute
    np.testing.assert_allclose(out[:, 3:nf], ref[:, 3:nf], atol=1e-5, rtol=0)


@pytest.mark.parametrize("cfg,n", [("C1", 100003), ("C2", 65536 + 17), ("C4", 32768 + 5)])
@pytest.mark.parametrize("frac", ["reference", "trilinear"])
def test_encode_fwd_vs_oracle_large(cfg, n, frac):
    emb, table, B = _embedder(cfg, 77, 0.5, frac)
    L, T, b, d = P.CONFIGS[cfg]
    x = P.make_points(5, n, -1.3, 1.3)
    with torch.no_grad():
        out = emb(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = O.encode_fwd(O.Grid(L, T, b, d), x, table, B, 0 if frac == "reference" else 1)
    nf = 3 + 2 * L
    if frac == "reference":
        assert np.array_equal(out[:, nf:], ref[:, nf:])
    else:
        np.testing.assert_allclose(out[:, nf:], ref[:, nf:], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[:, :nf], ref[:, :nf], atol=1e-5, rtol=0)


def test_encode_empty_and_ragged():
    emb, table, B = _embedder("tiny", 3, 0.5)
    with torch.no_grad():
        assert emb(torch.zeros((0, 3), device="cuda")).shape == (0, emb.embeddings_dim)
        for n in (1, 63, 64, 65, 129):
            x = P.make_points(n, n)
            out = emb(torch.from_numpy(x).cuda()).cpu().numpy()
            L, T, b, d = P.CONFIGS["tiny"]
            ref = O.encode_fwd(O.Grid(L, T, b, d), x, table, B, 0)
            assert np.array_equal(out[:, 3 + 2 * L:], ref[:, 3 + 2 * L:])
        # leading batch dims are preserved
        x = torch.from_numpy(P.make_points(9, 6)).cuda().reshape(2, 3, 3)
        assert emb(x).shape == (2, 3, emb.embeddings_dim)


@pytest.mark.parametrize("cfg", ["C1", "tiny"])
def test_encode_bwd_table_golden(golden, cfg):
    g = golden(f"encode_bwd_{cfg}")
    emb, _, _ = _embedder(cfg, int(g["seed"]), 0.5)
    x = torch.from_numpy(g["x"]).cuda()
    y = emb(x)
    (y * torch.from_numpy(g["d_out"]).cuda()).sum().backward()
    gt = emb.table.grad.cpu().numpy()
    nz = np.nonzero(np.abs(gt).sum(1))[0]
    assert np.array_equal(nz, g["nz_rows"])
    # (fp32 atomics in arrival order: rows of the tiny table sum ~100 terms of magnitude 1; 1.07e-6 seen on one element)
    np.testing.assert_allclose(gt[nz], g["nz_grad"], rtol=1e-5, atol=4e-6)


def test_encode_double_backward_linear_in_table():
    """The scatter's backward is the gather: check <scatter(d), T> == <d, gather(T)>."""
    from hashmodnffbanks_idr_amd import ops
    emb, _, _ = _embedder("tiny", 8, 0.5)
    x = torch.from_numpy(P.make_points(1, 500)).cuda()
    d = torch.randn(500, emb.n_levels * 2, device="cuda", requires_grad=True)
    Tt = torch.randn_like(emb.table)
    s = ops._HashScatter.apply(x, d, emb.desc, 0)
    (s * Tt).sum().backward()
    gathered = ops.encode_fwd(emb.desc, x, Tt, None, 0, hash_only=True)
    np.testing.assert_allclose(d.grad.cpu().numpy(), gathered.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_cpu_tensor_fails_loudly():
    from hashmodnffbanks_idr_amd._lib import HashmodError
    emb, _, _ = _embedder("tiny", 3, 0.5)
    with pytest.raises(HashmodError):
        emb(torch.zeros(4, 3))


@pytest.mark.parametrize("cfg", ["C2", "C4"])
def test_encode_full_size_properties(cfg):
    """BASELINE-size launch (2^22 points, the bench's gather workload) checked through properties that do not
    need the oracle to run 4 M points: (a) a 4096-row sample equals the oracle bit for bit (hash features),
    (b) permuting the points permutes the rows, bit for bit, (c) the hash features are linear in the table:
    doubling the table doubles them exactly (power of two), (d) table backward is the exact adjoint of the
    forward: <d_out, encode(T)> = <T, encode_bwd(d_out)> for the hash columns."""
    from hashmodnffbanks_idr_amd import ops
    n = 1 << 22
    emb, table, B = _embedder(cfg, 31, 0.5)
    L, T, b, d = P.CONFIGS[cfg]
    nf = 3 + 2 * L
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
    tab, Bf = emb.table.detach(), emb.freq_encoding.B
    out = ops.encode_fwd(emb.desc, x, tab, Bf, 0)
    idx = torch.randint(0, n, (4096,), generator=g)
    ref = O.encode_fwd(O.Grid(L, T, b, d), x[idx.cuda()].cpu().numpy(), table, B, 0)
    got = out[idx.cuda()].cpu().numpy()
    assert np.array_equal(got[:, nf:], ref[:, nf:])
    np.testing.assert_allclose(got[:, :nf], ref[:, :nf], atol=1e-5, rtol=0)
    perm = torch.randperm(n, generator=g).cuda()
    out_p = ops.encode_fwd(emb.desc, x[perm].contiguous(), tab, Bf, 0)
    assert torch.equal(out_p, out[perm])
    del out_p
    out2 = ops.encode_fwd(emb.desc, x, (tab * 2).contiguous(), Bf, 0)
    assert torch.equal(out2[:, nf:], out[:, nf:] * 2)
    assert torch.equal(out2[:, :nf], out[:, :nf])
    del out2
    m = 1 << 18                                        # adjoint identity on a 262 144-point slice (fp64 sums)
    d_feat = torch.randn((m, L * 2), generator=g).cuda()
    d_table = ops.encode_bwd_table(emb.desc, x[:m].contiguous(), d_feat, 0)
    lhs = (out[:m, nf:].double() * d_feat.double()).sum().item()
    rhs = (tab.double() * d_table.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs)), (lhs, rhs)


@pytest.mark.parametrize("mode", ["reference", "trilinear"])
def test_deterministic_table_backward(mode):
    """sort + segmented sum (hm_encode_rows / hm_encode_bwd_table_sorted) == the atomic scatter up to fp32 ordering,
    and bitwise identical from run to run (many points share rows: coarse levels, duplicated points)."""
    import params as P
    from hashmodnffbanks_idr_amd import ops
    L, T, b, d = P.CONFIGS["C1"]
    res, rows = P.level_table(L, T, b, d)
    desc = ops.GridDesc(res, rows, 2)
    x = torch.from_numpy(P.make_points(5, 20000, -1.0, 1.0)).cuda()
    x[5000:9000] = x[:4000]
    g = torch.randn(20000, L * 2, device="cuda")
    fm = ops.FRAC_MODES[mode]
    a = ops.encode_bwd_table(desc, x, g, fm)
    d1 = ops.encode_bwd_table(desc, x, g, fm, deterministic=True)
    d2 = ops.encode_bwd_table(desc, x, g, fm, deterministic=True)
    assert torch.equal(d1, d2)
    scale = a.abs().max().item()
    assert (a - d1).abs().max().item() <= 2e-5 * scale
    acc = torch.ones_like(d1)
    ops.encode_bwd_table(desc, x, g, fm, out=acc, deterministic=True)          # accumulates into the given tensor
    assert torch.allclose(acc - 1.0, d1, rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("mode", ["reference", "trilinear"])
@pytest.mark.parametrize("cfg", ["C2", "C4"])
def test_zordered_table_backward(cfg, mode):
    """big launches take the z-ordered, LDS-privatised scatter (hm_encode_bwd_table_ws): same gradient as the atomic
    kernel on the same points (fp32 summation order differs), nothing lost on ragged sizes or outside [-1,1]^3"""
    import params as P
    from hashmodnffbanks_idr_amd import _lib, ops
    L, T, b, d = P.CONFIGS[cfg]
    res, rows = P.level_table(L, T, b, d)
    desc = ops.GridDesc(res, rows, 2)
    n = 200000 + 37
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.rand((n, 3), generator=g) * 2.6 - 1.3).cuda()          # some points outside the unit cube
    x[2000:4000] = x[:2000].clone()
    gf = torch.randn((n, L * 2), generator=g).cuda()
    fm = ops.FRAC_MODES[mode]
    got = ops.encode_bwd_table(desc, x, gf, fm)                        # n >= 131072 -> workspace path
    ref = torch.zeros_like(got)
    _lib.check(_lib.lib().hm_encode_bwd_table(desc.handle, _lib.dptr(x), n, _lib.dptr(gf), gf.stride(0), _lib.dptr(ref),
                                              fm, _lib.stream_ptr(x)))
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print(f"z-ordered table backward {cfg}/{mode}: max |d| {err:.3e} (scale {scale:.3e})")
    assert err <= 3e-5 * scale
    assert abs(got.double().sum().item() - ref.double().sum().item()) <= 1e-5 * ref.double().abs().sum().item()


@pytest.mark.parametrize("frac", ["reference", "trilinear"])
@pytest.mark.parametrize("cfg,n", [("C2", 131072 + 37), ("C2", 200037), ("C4", 131072 + 37), ("C4", 200037)])
def test_zordered_forward_vs_oracle(cfg, n, frac):
    """launches >= 131 072 points over tables > 8 MiB take the z-ordered gather (hm_encode_fwd_ws: the kernel the bench's
    `roofline` object times): ragged last tile, points outside [-1, 1]^3 (the z-slab clamp, hm_encode.hip z_slab) and
    duplicated points, both weight modes, every row against the C oracle - hash columns bit for bit in reference mode
    (hashGridEmbedding.py:81-102)"""
    from hashmodnffbanks_idr_amd import _lib, ops
    emb, table, B = _embedder(cfg, 78, 0.5, frac)
    L, T, b, d = P.CONFIGS[cfg]
    x = P.make_points(6, n, -1.3, 1.3)
    x[3000:5000] = x[:2000]
    x[7] = (1.0, -1.0, 1.0)
    x[8] = (-1.0, 1.0, -1.0)
    fm = ops.FRAC_MODES[frac]
    xt = torch.from_numpy(x).cuda()
    need = _lib.check(_lib.lib().hm_encode_workspace_bytes(emb.desc.handle, n))
    assert need > 0
    out = ops.encode_fwd(emb.desc, xt, emb.table.detach(), emb.freq_encoding.B, fm)
    # the same launch through the tile kernel (no workspace): a second implementation of the same rows
    tile = torch.empty_like(out)
    _lib.check(_lib.lib().hm_encode_fwd(emb.desc.handle, _lib.dptr(xt), n, _lib.dptr(emb.table.detach()),
                                        _lib.dptr(emb.freq_encoding.B), _lib.dptr(tile), emb.desc.E, fm,
                                        _lib.stream_ptr(xt)))
    out, tile = out.cpu().numpy(), tile.cpu().numpy()
    ref = O.encode_fwd(O.Grid(L, T, b, d), x, table, B, fm)
    nf = 3 + 2 * L
    if frac == "reference":
        assert np.array_equal(out[:, nf:], ref[:, nf:]), "z-ordered gather: hash columns differ from the oracle"
        assert np.array_equal(tile[:, nf:], ref[:, nf:])
    else:
        err = np.abs(out[:, nf:] - ref[:, nf:]).max()
        print(f"z-ordered gather {cfg}/trilinear n={n}: max |d| vs the oracle {err:.3e}")
        np.testing.assert_allclose(out[:, nf:], ref[:, nf:], rtol=1e-5, atol=1e-6)
    assert np.array_equal(out[:, :3], x)
    np.testing.assert_allclose(out[:, :nf], ref[:, :nf], atol=1e-5, rtol=0)
    assert np.array_equal(out, tile), "z-ordered and tile kernels must produce identical rows"


@pytest.mark.parametrize("mode", ["reference", "trilinear"])
@pytest.mark.parametrize("cfg", ["C2", "C4"])
def test_zordered_table_backward_vs_oracle(cfg, mode):
    """the z-ordered scatter against the C oracle's table backward (not only against the repo's own atomic kernel):
    ragged size, points outside the cube, duplicates"""
    from hashmodnffbanks_idr_amd import ops
    L, T, b, d = P.CONFIGS[cfg]
    res, rows = P.level_table(L, T, b, d)
    desc = ops.GridDesc(res, rows, 2)
    n = 200000 + 37
    x = P.make_points(9, n, -1.3, 1.3)
    x[2000:4000] = x[:2000]
    rs = np.random.RandomState(10)
    gf = rs.standard_normal((n, L * 2)).astype(np.float32)
    fm = ops.FRAC_MODES[mode]
    got = ops.encode_bwd_table(desc, torch.from_numpy(x).cuda(), torch.from_numpy(gf).cuda(), fm).cpu().numpy()
    d_out = np.zeros((n, 3 + 2 * L + 2 * L), np.float32)
    d_out[:, 3 + 2 * L:] = gf
    ref = O.encode_bwd_table(O.Grid(L, T, b, d), x, d_out, fm)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    print(f"z-ordered table backward vs the C oracle {cfg}/{mode}: max |d| {err:.3e} (scale {scale:.3e})")
    assert err <= 3e-5 * scale
    assert abs(got.astype(np.float64).sum() - ref.astype(np.float64).sum()) <= 1e-5 * np.abs(ref).astype(np.float64).sum()


def _trilinear_torch(x, table, desc):
    """differentiable torch restatement of the trilinear encoder (fp64): floor voxel, 8 hashed corners, product
    weights - the opt-in mode has no counterpart in the reference, so autograd on this expression is the checker"""
    feats = []
    xd = x.double()
    for l in range(desc.L):
        r, rows, off = int(desc.res[l]), int(desc.rows[l]), int(desc.row_off[l])
        xs = (x.detach() * float(r)).double()              # value: the kernel forms x * res in fp32
        xs = xs + (xd * float(r) - (xd * float(r)).detach())   # gradient: of the exact product
        fl = torch.floor(xs.detach())
        t = xs - fl
        acc = 0.0
        for c in range(8):
            w, h = 1.0, None
            for d, prime in enumerate((1, 3, 2654435761)):
                bit = (c >> d) & 1
                w = w * (t[:, d] if bit else 1.0 - t[:, d])
                u = ((fl[:, d].long() + bit) & 0xFFFFFFFF) * prime & 0xFFFFFFFF
                h = u if h is None else h ^ u
            acc = acc + w[:, None] * table[off + (h % rows)].double()
        feats.append(acc)
    return torch.cat(feats, 1)


@pytest.mark.parametrize("cfg", ["C1", "C2"])
def test_trilinear_input_gradient_first_and_second_order(cfg):
    """frac_mode='trilinear': d(features)/dx (hm_encode_bwd_input) and the three backward products of that gradient
    (hm_encode_jvp, hm_encode_bwd_table_jvp, second-order hm_encode_bwd_input) against torch autograd on the fp64
    restatement - the pattern ImplicitNetwork.gradient(create_graph=True) + loss.backward() produces"""
    import params as P
    from hashmodnffbanks_idr_amd import ops
    L, T, b, d = P.CONFIGS[cfg]
    res, rows = P.level_table(L, T, b, d)
    desc = ops.GridDesc(res, rows, 2)
    n = 3000
    g = torch.Generator(device="cpu").manual_seed(11)
    x0 = (torch.rand((n, 3), generator=g) * 2.2 - 1.1).cuda()
    tab0 = (torch.rand((desc.total_rows, 2), generator=g) - 0.5).cuda()
    m_out = torch.randn((n, L * 2), generator=g).cuda()       # stands for the MLP between features and sdf
    r_vec = torch.randn((n, 3), generator=g).cuda()

    def run(features):
        x = x0.clone().requires_grad_(True)
        tab = tab0.clone().requires_grad_(True)
        m = m_out.clone().requires_grad_(True)
        e = features(x, tab)
        y = (e * m.to(e.dtype)).sum()
        (gx,) = torch.autograd.grad(y, x, create_graph=True)
        loss = (gx * r_vec.to(gx.dtype)).pow(2).sum() + 0.1 * e.pow(2).sum()
        loss.backward()
        return e.detach(), gx.detach(), x.grad, tab.grad, m.grad

    got = run(lambda x, tab: ops.hash_features(x, tab, desc, ops.FRAC_MODES["trilinear"]))
    ref = run(lambda x, tab: _trilinear_torch(x, tab, desc))
    for name, a, b_ in zip(("features", "d/dx", "loss d/dx (second order)", "loss d/dtable", "loss d/d(d_feat)"), got, ref):
        scale = b_.abs().max().item()
        err = (a.double() - b_.double()).abs().max().item()
        print(f"trilinear {cfg} {name}: max |d| {err:.3e} (scale {scale:.3e})")
        assert scale > 0 and err <= 2e-4 * scale, name


def test_trilinear_mode_trains_with_eikonal_term():
    """IDRNetwork with the trilinear encoder takes optimizer steps (eikonal + normals differentiate the encoder twice)"""
    from helpers import idr_conf
    import bench
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    torch.manual_seed(0)
    conf = idr_conf("C1")
    model = IDRNetwork(conf).cuda()
    emb = model.implicit_network.embed_model.embedder_obj
    emb.frac_mode = "trilinear"
    with torch.no_grad():
        model.implicit_network.lin0.weight_v[:, 3:].normal_(0, 0.02)
        emb.table.uniform_(-0.05, 0.05)
    model.train()
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    inp, gt = bench.synthetic_batch(7, 512, "cuda")
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = model(inp)
        lo = loss_fn(out, gt)
        lo["loss"].backward()
        assert emb.table.grad is not None and torch.isfinite(emb.table.grad).all()
        assert emb.table.grad.abs().sum().item() > 0
        opt.step()
        losses.append(lo["loss"].item())
    assert all(np.isfinite(losses)), losses


@pytest.mark.parametrize("cfg", ["C1", "C2"])
def test_embedding_row_input_gradient_matches_torch_autograd(cfg):
    """MultiResHashGridMLP with points that need a gradient: the fused route (one encoder launch + the hand-written
    first / second order kernels of the Fourier columns, hm_fourier_bwd_input[_bwd]) against torch's autograd over the
    FourierFeature expression (frequency_enc.py:63-67) - values, d/dx with create_graph, and the backward of a loss
    on that gradient w.r.t. x, the upstream gradient and the table"""
    import params as P
    from hashmodnffbanks_idr_amd.model.embeddings.hashGridEmbedding import MultiResHashGridMLP
    L, T, b, d = P.CONFIGS[cfg]
    torch.manual_seed(0)
    emb = MultiResHashGridMLP(True, 3, L, 2, T, b, d).cuda()
    with torch.no_grad():
        emb.table.uniform_(-0.5, 0.5)
    n = 3000
    g = torch.Generator(device="cpu").manual_seed(5)
    x0 = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
    m0 = torch.randn((n, emb.embeddings_dim), generator=g).cuda()
    r_vec = torch.randn((n, 3), generator=g).cuda()

    def run(fused):
        emb.fused_input_grad = fused
        emb.table.grad = None
        x = x0.clone().requires_grad_(True)
        m = m0.clone().requires_grad_(True)
        e = emb(x)
        (gx,) = torch.autograd.grad((e * m).sum(), x, create_graph=True)
        assert emb.table.grad is None                      # the create_graph pass must not touch the table
        loss = (gx * r_vec).pow(2).sum() + 0.1 * e.pow(2).sum()
        loss.backward()
        return e.detach(), gx.detach(), x.grad, m.grad, emb.table.grad.clone()

    got, ref = run(True), run(False)
    for name, a, b_ in zip(("row", "d/dx", "loss d/dx", "loss d/dm", "loss d/dtable"), got, ref):
        scale = b_.abs().max().item()
        err = (a - b_).abs().max().item()
        print(f"embedding row {cfg} {name}: max |d| {err:.3e} (scale {scale:.3e})")
        assert scale > 0 and err <= 2e-5 * scale, name


@pytest.mark.parametrize("n,bits", [(1, 5), (255, 8), (4096, 12), (4097, 23), (100003, 23), (300000, 25), (70001, 31)])
def test_own_radix_sort_equals_stable_sort(n, bits):
    """hm_sort_pairs_i32 (csrc/hm_sort.hip) against torch's stable sort: same sorted keys AND the same permutation (equal
    keys keep their input order) - ragged sizes, every pass count (1 .. 4), heavy duplication"""
    from hashmodnffbanks_idr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n)
    hi = min((1 << bits) - 1, 2 ** 31 - 1)
    keys = torch.randint(0, hi + 1, (n,), generator=g, dtype=torch.int64)
    keys[: n // 3] = keys[: n // 3] % 17                  # many equal keys
    if n > 10:
        keys[-5:] = hi                                    # the largest key of the range
    keys = keys.to(torch.int32).cuda()
    sk, perm = ops.sort_pairs(keys, bits)
    rk, rp = torch.sort(keys, stable=True)
    assert torch.equal(sk, rk)
    assert torch.equal(perm, rp)
    assert ops.sort_pairs(keys[:0], bits)[0].numel() == 0


@pytest.mark.parametrize("cfg", ["viewdir", "tiny"])
@pytest.mark.parametrize("mode", ["reference", "trilinear"])
def test_encode_bwd_table_small_table_path(cfg, mode):
    """tables of <= 4096 rows with many contributions take the LDS-privatised scatter (encode_bwd_table_small_kernel:
    the view-direction grid of the rendering network has 8 rows per level, 2048 rays land on them): against the C
    oracle's scatter and - adjoint identity - against the forward gather, in both weight modes, accumulating into a
    non-zero tensor."""
    from hashmodnffbanks_idr_amd import ops
    emb, table, B = _embedder(cfg, 5, 0.5, frac_mode=mode)
    B = torch.from_numpy(B).cuda()
    L, T, b, d = P.CONFIGS[cfg]
    assert emb.desc.total_rows <= 4096
    fm = ops.FRAC_MODES[mode]
    n = 3001
    g = torch.Generator(device="cpu").manual_seed(21)
    x = (torch.rand((n, 3), generator=g) * 2.2 - 1.1)
    d_feat = torch.randn((n, L * 2), generator=g)
    got = ops.encode_bwd_table(emb.desc, x.cuda(), d_feat.cuda(), fm)
    d_out = np.zeros((n, 3 + 2 * L + 2 * L), np.float32)
    d_out[:, 3 + 2 * L:] = d_feat.numpy()
    ref = O.encode_bwd_table(O.Grid(L, T, b, d), x.numpy(), d_out, fm)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-6 * scale)
    feats = ops.encode_fwd(emb.desc, x.cuda(), emb.table.detach(), B, fm, hash_only=True)
    lhs = (feats.double() * d_feat.cuda().double()).sum().item()
    rhs = (emb.table.detach().double() * got.double()).sum().item()
    mag = (feats.double().abs() * d_feat.cuda().double().abs()).sum().item()     # (the sum cancels: ~3000 fp32 terms per row)
    assert abs(lhs - rhs) <= 1e-6 * mag, (lhs, rhs, mag)
    acc = torch.full_like(got, 0.25)
    ops.encode_bwd_table(emb.desc, x.cuda(), d_feat.cuda(), fm, out=acc)
    np.testing.assert_allclose((acc - 0.25).cpu().numpy(), ref, rtol=0, atol=4e-6 * scale)
