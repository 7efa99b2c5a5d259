"""hm_gemm_f32_ep: GEMM with the fused Softplus epilogues of the SDF MLP's gradient sweeps, against the same
arithmetic composed from torch ops in float64 (model/implicit_differentiable_renderer.py:84,102,116-128)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BETA, THR = 100.0, 20.0


def _s1s2(z):
    bz = z * BETA
    e = torch.exp(torch.clamp(bz, max=80.0))
    s1 = torch.where(bz > THR, torch.ones_like(z), e / (e + 1))
    s2 = torch.where(bz > THR, torch.zeros_like(z), BETA * e / (e + 1) ** 2)
    return s1, s2


def _close(got, ref, what):
    # fp32 accumulation error scales with the magnitude of the products (s2 reaches beta/4 = 25)
    atol = max(2e-5, 2e-6 * float(ref.abs().max()))
    np.testing.assert_allclose(got.double().cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=atol, err_msg=what)


# (3072 / 2700 rows: the 96 x 64 tile of the pipelined kernel - whole and with a ragged last row tile, 445 = ragged columns)
# (K = 445 / 257: the pipelined kernel's K tail - the k range runs to the next multiple of 128, the operand with k as its
#  slow dimension returns zeros beyond its end - in the NN form of the backward sweeps; their NT forms stay on the generic kernel)
@pytest.mark.parametrize("M,N,K", [(300, 445, 67), (2100, 512, 512), (64, 257, 512), (1, 5, 3), (3072, 512, 512),
                                   (2700, 445, 512), (3072, 512, 445), (2048, 512, 257), (100, 130, 190)])
def test_gemm_epilogues(M, N, K):
    from hashmodnffbanks_idr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.3).cuda()
    w = (torch.randn(N, K, generator=g) * 0.2).cuda()          # nn.Linear weight [out, in]
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    wide = (torch.randn(M, N + 9, generator=g) * 0.05).cuda()   # z / g are column views of wider tensors
    z, gg = wide[:, 3:3 + N], (torch.randn(M, N + 4, generator=g)).cuda()[:, 4:]
    a64, w64, b64, z64, g64 = a.double(), w.double(), bias.double(), z.double(), gg.double()
    v64 = a64 @ w64.t() + b64

    c, h = ops.gemm_ep(a, w, bias, False, True, ops.EPI_SOFTPLUS, BETA, THR)
    _close(c, v64, "softplus: raw")
    _close(h, torch.nn.functional.softplus(v64, beta=BETA, threshold=THR), "softplus: activation")

    s1, s2 = _s1s2(z64)
    sc = 0.7071067811865476
    nz = max(1, N - 7)
    c, o = ops.gemm_ep(a, w, None, False, True, ops.EPI_S1MUL, BETA, THR, scale=sc, z=z, g=gg, nz=nz)
    raw = (a64 @ w64.t()) * sc
    _close(c, raw, "s1mul: raw")
    assert o.shape == (M, nz)
    _close(o, raw[:, :nz] * s1[:, :nz] + g64[:, :nz], "s1mul: product + addend")
    c, o = ops.gemm_ep(a, w, None, False, True, ops.EPI_S1MUL, BETA, THR, z=z, want_c=False)
    assert c is None
    _close(o, (a64 @ w64.t()) * s1, "s1mul: no raw, no addend")

    o1, o2, o3 = ops.gemm_ep(a, w, None, False, True, ops.EPI_ADJOINT, BETA, THR, z=z, g=gg, want_out3=True)
    ub = a64 @ w64.t()
    _close(o1, ub * s1, "adjoint: out1")
    _close(o2, ub * g64 * s2, "adjoint: out2")
    _close(o3, g64 * s1, "adjoint: out3")
    # NN form (dY * W) as the backward sweeps use it
    wt = w.t().contiguous()                                     # [K, N]
    c, o = ops.gemm_ep(a, wt, None, False, False, ops.EPI_S1MUL, BETA, THR, z=z)
    _close(o, (a64 @ wt.double()) * s1, "s1mul NN")


def test_weight_norm_fold_multi_matches_torch():
    """ops.weight_norm_fold: W = g v / ||v|| for a list of layers in one launch and its backward in one launch, against
    torch._weight_norm per layer (what nn.utils.weight_norm runs, implicit_differentiable_renderer.py:95-97)"""
    from hashmodnffbanks_idr_amd import ops
    g0 = torch.Generator(device="cpu").manual_seed(0)
    shapes = [(512, 67), (512, 512), (445, 512), (512, 512), (257, 512), (3, 512), (1, 7)]
    vs = [torch.randn(s, generator=g0).cuda().requires_grad_(True) for s in shapes]
    gs = [(torch.rand((s[0], 1), generator=g0) + 0.5).cuda().requires_grad_(True) for s in shapes]
    ms = [torch.randn(s, generator=g0).cuda() for s in shapes]
    ws = ops.weight_norm_fold(vs, gs)
    sum((w * m).sum() for w, m in zip(ws, ms)).backward()
    got = [(w.detach(), v.grad.clone(), g.grad.clone()) for w, v, g in zip(ws, vs, gs)]
    for v, g in zip(vs, gs):
        v.grad = g.grad = None
    ref_w = [torch._weight_norm(v, g, 0) for v, g in zip(vs, gs)]
    sum((w * m).sum() for w, m in zip(ref_w, ms)).backward()
    for (w, gv, gg), rw, v, g in zip(got, ref_w, vs, gs):
        assert gg.shape == g.grad.shape
        for a, b in ((w, rw.detach()), (gv, v.grad), (gg, g.grad)):
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-6)


def test_grouped_weight_gradient_gemm_matches_torch():
    """hm_gemm_f32_group_tn: C_p += A_p^T B_p for a whole list in one launch - the layer shapes of the SDF / rendering
    networks (partial edge tiles: 445, 257, 67, 281 columns), strided operand views, a K that is not a multiple of 128
    (falls back to hm_gemm_f32 for that item), more items than one launch holds, accumulation into non-zero C"""
    from hashmodnffbanks_idr_amd import ops
    rs = np.random.RandomState(5)
    T = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32)).cuda()   # noqa: E731
    shapes = [(6144, 512, 512), (6144, 445, 512), (6144, 512, 67), (3072, 257, 512), (4096, 512, 512), (2048, 512, 281),
              (2048, 3, 512), (1000, 64, 96), (128, 65, 33)] + [(1024, 130, 70)] * 12          # 21 items > one launch
    probs, refs = [], []
    for (K, M, N) in shapes:
        a_full, b_full = T(K, M + 5), T(K, N + 3)
        a, b = a_full[:, 2:2 + M], b_full[:, :N]                      # views with row strides M + 5 / N + 3
        c = T(M, N)
        refs.append((c.double() + a.double().t() @ b.double()).float())
        probs.append((a, b, c))
    ops.gemm_group_tn(probs)
    torch.cuda.synchronize()
    worst = 0.0
    for (a, b, c), ref, (K, M, N) in zip(probs, refs, shapes):
        err = (c - ref).abs().max().item() / (np.sqrt(K) + 1e-30)
        worst = max(worst, err)
        np.testing.assert_allclose(c.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-5 * np.sqrt(K))
    print(f"grouped weight-gradient GEMM: worst |d| / sqrt(K) {worst:.3e}")
    ops.gemm_group_tn([])                                             # empty list is a no-op
    with pytest.raises(ValueError):
        ops.gemm_group_tn([(T(8, 4), T(9, 4), T(4, 4))])              # inner dimensions differ
