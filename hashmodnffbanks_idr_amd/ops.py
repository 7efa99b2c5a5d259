"""Thin Python operators over the C ABI (include/hashmod.h) + the autograd glue.

Every function here launches hand-written HIP kernels from libhashmod.so on torch's current
stream; torch supplies device memory only.  CPU tensors raise (no fallback).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, dptr, lib, require_gpu, stream_ptr

FRAC_MODES = {"reference": 0, "trilinear": 1}


def level_table(n_levels, log2_hashmap_size, base_resolution, desired_resolution, in_dim=3):
    """res[l], rows[l] in host double precision, the arithmetic of
    reference model/embeddings/hashGridEmbedding.py:126-132 (Python ``math``, not fp32)."""
    beta_growth = math.exp((math.log(desired_resolution) - math.log(base_resolution)) / (n_levels - 1))
    res, rows = [], []
    for level_idx in range(n_levels):
        resolution = math.floor(base_resolution * (beta_growth ** level_idx))
        res.append(resolution)
        rows.append(min(resolution ** in_dim, 2 ** log2_hashmap_size))
    return res, rows


class GridDesc:
    """Owns an hm_grid_desc (host-side level table handed to kernels by value)."""

    def __init__(self, res, rows, n_features=2):
        self.L = len(res)
        self.F = int(n_features)
        self.res = np.asarray(res, np.int32)
        self.rows = np.asarray(rows, np.uint32)
        self.row_off = np.concatenate([[0], np.cumsum(self.rows.astype(np.uint64))]).astype(np.uint64)
        self.total_rows = int(self.row_off[-1])
        self.E = 3 + 2 * self.L + self.L * self.F
        self._h = C.c_void_p(0)
        check(lib().hm_grid_desc_create(self.L, self.F, self.res.ctypes.data_as(C.c_void_p),
                                        self.rows.ctypes.data_as(C.c_void_p),
                                        self.row_off.ctypes.data_as(C.c_void_p), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def __reduce__(self):  # copy.deepcopy / pickle: rebuild the native descriptor from the host arrays
        return (GridDesc, (self.res.tolist(), self.rows.tolist(), self.F))

    def __del__(self):
        try:
            if self._h:
                lib().hm_grid_desc_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass


def _prep_x(x):
    if x.dtype != torch.float32:
        x = x.float()
    return x.reshape(-1, 3).contiguous()


def corner_ids(desc, level, x):
    """xi [N,3] int32 and the 8 corner row ids [N,8] (as int64) of one level - bit-exact with
    reference hash_func (hashGridEmbedding.py:32-40,84-98)."""
    x = _prep_x(x)
    require_gpu(x)
    n = x.shape[0]
    xi = torch.empty((n, 3), dtype=torch.int32, device=x.device)
    ids = torch.empty((n, 8), dtype=torch.int32, device=x.device)
    check(lib().hm_corner_ids(desc.handle, int(level), dptr(x), n, dptr(xi), dptr(ids), stream_ptr(x)))
    return xi, ids.long() & 0xFFFFFFFF


_ENC_BWD_WS = {}   # device -> scratch of the z-ordered table backward
_ENC_WS = {}   # device -> scratch of the z-ordered encode (grown on demand, reused by every launch on that stream)


def encode_fwd(desc, x, table, B, frac_mode=0, hash_only=False):
    """[N,E] embedding (or [N,L*F] hash features when hash_only) - no autograd."""
    x = _prep_x(x)
    require_gpu(x, table, B)
    assert table.is_contiguous() and table.dtype == torch.float32 and table.shape == (desc.total_rows, desc.F)
    n = x.shape[0]
    width = desc.L * desc.F if hash_only else desc.E
    Bp = None if hash_only else B.contiguous()
    if n >= 131072:     # big launches: z-ordered gather (needs scratch for the point permutation)
        # rows padded to a multiple of 4 floats (268 -> 272 bytes at E = 67): every row then starts on a 16-byte boundary
        # and leaves the kernel as ONE dwordx4 store instruction; the caller gets the [n, width] view of the buffer
        stride = (width + 3) & ~3
        buf = torch.empty((n, stride), dtype=torch.float32, device=x.device)
        out = buf[:, :width] if stride != width else buf
        need = check(lib().hm_encode_workspace_bytes(desc.handle, n))
        ws = _ENC_WS.get(x.device)
        if ws is None or ws.numel() < need:
            ws = _ENC_WS[x.device] = torch.empty(need, dtype=torch.uint8, device=x.device)
        check(lib().hm_encode_fwd_ws(desc.handle, dptr(x), n, dptr(table), dptr(Bp), dptr(buf), stride, int(frac_mode),
                                     dptr(ws), ws.numel(), stream_ptr(x)))
    else:
        out = torch.empty((n, width), dtype=torch.float32, device=x.device)
        check(lib().hm_encode_fwd(desc.handle, dptr(x), n, dptr(table), dptr(Bp), dptr(out), width, int(frac_mode),
                                  stream_ptr(x)))
    return out


def encode_bwd_table(desc, x, d_feat, frac_mode=0, out=None, deterministic=False):
    """Scatter-add of the hash-feature gradient d_feat [N,L*F] into a [rows,F] table gradient.
    deterministic: sort the contributions by destination row (sort_pairs: the library's stable radix sort) and sum each row's run in one thread
    (hm_encode_rows + hm_encode_bwd_table_sorted) instead of fp32 atomics - bitwise reproducible."""
    x = _prep_x(x)
    require_gpu(x, d_feat)
    n = x.shape[0]
    assert d_feat.shape == (n, desc.L * desc.F) and d_feat.dtype == torch.float32
    if d_feat.stride(1) != 1:
        d_feat = d_feat.contiguous()
    if out is None:
        out = torch.zeros((desc.total_rows, desc.F), dtype=torch.float32, device=x.device)
    if deterministic and n > 0:
        corners = 1 if int(frac_mode) == 0 else 8
        keys = torch.empty(n * desc.L * corners, dtype=torch.int32, device=x.device)
        wts = torch.empty(n * desc.L * corners, dtype=torch.float32, device=x.device) if corners == 8 else None
        check(lib().hm_encode_rows(desc.handle, dptr(x), n, int(frac_mode), dptr(keys), dptr(wts), stream_ptr(x)))
        skeys, perm = sort_pairs(keys, max(int(desc.total_rows - 1).bit_length(), 1))
        check(lib().hm_encode_bwd_table_sorted(desc.handle, dptr(skeys), dptr(perm), keys.numel(), corners, dptr(d_feat),
                                               d_feat.stride(0), dptr(wts), dptr(out), stream_ptr(x)))
        return out
    if n >= 131072 and desc.F == 2:     # big launches: z-ordered, LDS-privatised scatter (needs scratch)
        need = check(lib().hm_encode_bwd_workspace_bytes(desc.handle, n))
        ws = _ENC_BWD_WS.get(x.device)
        if ws is None or ws.numel() < need:
            ws = _ENC_BWD_WS[x.device] = torch.empty(need, dtype=torch.uint8, device=x.device)
        check(lib().hm_encode_bwd_table_ws(desc.handle, dptr(x), n, dptr(d_feat), d_feat.stride(0), dptr(out),
                                           int(frac_mode), dptr(ws), ws.numel(), stream_ptr(x)))
        return out
    check(lib().hm_encode_bwd_table(desc.handle, dptr(x), n, dptr(d_feat), d_feat.stride(0), dptr(out),
                                    int(frac_mode), stream_ptr(x)))
    return out


_SORT_WS = {}


def sort_pairs(keys, key_bits=31):
    """(sorted keys, permutation as int64) of non-negative int32 keys < 2^key_bits: the library's stable LSD radix sort
    (hm_sort_pairs_i32), the sort behind encode_bwd_table(deterministic=True)."""
    require_gpu(keys)
    if keys.dtype != torch.int32 or keys.dim() != 1 or not keys.is_contiguous():
        raise ValueError("hashmod sort_pairs: contiguous 1-D int32 keys expected")
    n = keys.numel()
    out = torch.empty_like(keys)
    perm = torch.empty(n, dtype=torch.int64, device=keys.device)
    if n == 0:
        return out, perm
    need = check(lib().hm_sort_workspace_bytes(n))
    ws = _SORT_WS.get(keys.device)
    if ws is None or ws.numel() < need:
        ws = _SORT_WS[keys.device] = torch.empty(need, dtype=torch.uint8, device=keys.device)
    check(lib().hm_sort_pairs_i32(dptr(keys), n, int(key_bits), dptr(out), dptr(perm), dptr(ws), ws.numel(),
                                  stream_ptr(keys)))
    return out, perm


class _HashFeatures(torch.autograd.Function):
    """Encoder output as an autograd node: hash features [N,L*F] (B is None) or the full
    [N,E] row [x|sin|cos|features] (B given; only the feature columns carry gradient).
    Differentiable to any order w.r.t. the table (the op is linear in it).  In reference frac mode
    d/dx of the hash features is identically zero, exactly as in the reference where
    xf = x - x.float() kills the interpolation weights (hashGridEmbedding.py:86); in trilinear mode
    d/dx is the node _HashInputGrad (differentiable once more)."""

    @staticmethod
    def forward(ctx, x, table, B, desc, frac_mode, collector=None):
        ctx.desc, ctx.frac_mode, ctx.collector = desc, frac_mode, collector
        ctx.hoff = 0 if B is None else 3 + 2 * desc.L
        ctx.save_for_backward(x)
        ctx.table = table if (frac_mode != 0 and ctx.needs_input_grad[0]) else None   # d/dx gathers the table again
        return encode_fwd(desc, x, table, B, frac_mode, hash_only=B is None)

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        d_table = d_x = None
        if ctx.needs_input_grad[1]:
            d_feat = d_out[:, ctx.hoff:] if ctx.hoff else d_out
            if ctx.collector is not None and ctx.collector.active:
                # data-parallel static step (parallel.TouchedRowExchange): the collector scatters into ITS static dense
                # gradient (bound as table.grad) and lists the touched rows for the exchange; autograd gets nothing
                ctx.collector.add(x, d_feat)
                d_table = None
            else:
                d_table = _HashScatter.apply(x, d_feat, ctx.desc, ctx.frac_mode)
        if ctx.needs_input_grad[0]:
            if ctx.hoff:
                raise RuntimeError("internal: full-row encode node must not be used when x requires grad")
            if ctx.frac_mode != 0:
                d_x = _HashInputGrad.apply(x, ctx.table, d_out, ctx.desc)
        return d_x, d_table, None, None, None, None


class _HashInputGrad(torch.autograd.Function):
    """gx = J(x)^T d_feat of the trilinear encoder (J = d features / d x) as a node of its own, so that
    ImplicitNetwork.gradient(create_graph=True) can be differentiated once more: the eikonal / normal terms' backward
    arrives here as gg_x and leaves towards d_feat (J gg_x), the table and x (csrc/hm_encode_dx.hip)."""

    @staticmethod
    def forward(ctx, x, table, d_feat, desc):
        ctx.desc = desc
        d_feat = _rowmajor(d_feat)
        ctx.save_for_backward(x, table, d_feat)
        gx = torch.empty_like(x)
        check(lib().hm_encode_bwd_input(desc.handle, dptr(x), x.shape[0], dptr(table), dptr(d_feat), _ld(d_feat),
                                        None, dptr(gx), stream_ptr(x)))
        return gx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg_x):
        x, table, d_feat = ctx.saved_tensors
        desc, n = ctx.desc, x.shape[0]
        gg_x = gg_x.contiguous()
        d_x = d_table = d_dfeat = None
        if ctx.needs_input_grad[0]:
            d_x = torch.empty_like(x)
            check(lib().hm_encode_bwd_input(desc.handle, dptr(x), n, dptr(table), dptr(d_feat), _ld(d_feat),
                                            dptr(gg_x), dptr(d_x), stream_ptr(x)))
        if ctx.needs_input_grad[1]:
            d_table = torch.zeros_like(table)
            check(lib().hm_encode_bwd_table_jvp(desc.handle, dptr(x), n, dptr(gg_x), dptr(d_feat), _ld(d_feat),
                                                dptr(d_table), stream_ptr(x)))
        if ctx.needs_input_grad[2]:
            d_dfeat = torch.empty((n, desc.L * desc.F), dtype=torch.float32, device=x.device)
            check(lib().hm_encode_jvp(desc.handle, dptr(x), n, dptr(table), dptr(gg_x), dptr(d_dfeat),
                                      d_dfeat.stride(0), stream_ptr(x)))
        return d_x, d_table, d_dfeat, None


class _HashScatter(torch.autograd.Function):
    """d_table = scatter_add(d_feat at x); its own backward is the gather again."""

    @staticmethod
    def forward(ctx, x, d_feat, desc, frac_mode):
        ctx.desc, ctx.frac_mode = desc, frac_mode
        ctx.save_for_backward(x)
        return encode_bwd_table(desc, x, d_feat, frac_mode)

    @staticmethod
    def backward(ctx, gg_table):
        (x,) = ctx.saved_tensors
        gg = None
        if ctx.needs_input_grad[1]:
            gg = _HashFeatures.apply(x, gg_table.contiguous(), None, ctx.desc, ctx.frac_mode, None)
        return None, gg, None, None


class _AttachInputGrad(torch.autograd.Function):
    """Puts the full embedding row [x | sin | cos | hash features] (already computed by ONE encoder launch as a
    table-gradient-only node, encode_table_grad) onto the graph of the points x: the row is returned as it is
    (mark_dirty, no copy); backward hands d_row on to the table node and produces d_x = _RowInputGrad.  Two nodes, so
    that autograd.grad(e, x, create_graph=True) (ImplicitNetwork.gradient) does not run the table backward at all:
    the engine only visits producers of the inputs that were asked for.  Reference frac mode only (the hash features
    do not depend on x there, hashGridEmbedding.py:86)."""

    @staticmethod
    def forward(ctx, row, x, B):
        ctx.save_for_backward(x, B)
        ctx.mark_dirty(row)
        return row

    @staticmethod
    def backward(ctx, d_row):
        x, B = ctx.saved_tensors
        d_x = _RowInputGrad.apply(x, B, d_row) if ctx.needs_input_grad[1] else None
        return (d_row if ctx.needs_input_grad[0] else None), d_x, None


class _RowInputGrad(torch.autograd.Function):
    """gx = d_row[:, :3] + (d Fourier columns / d x)^T d_row as ONE kernel, differentiable once more: the eikonal /
    normal terms' backward arrives as gg and leaves towards d_row and x in one kernel too (csrc/hm_encode_dx.hip).
    The torch expression of the same thing is ~10 elementwise kernels and a K = 3 vendor matmul per pass."""

    @staticmethod
    def forward(ctx, x, B, d_row):
        d_row = _rowmajor(d_row)
        ctx.save_for_backward(x, B, d_row)
        gx = torch.empty_like(x)
        check(lib().hm_fourier_bwd_input(dptr(x), x.shape[0], dptr(B), B.shape[1], dptr(d_row), _ld(d_row), dptr(gx),
                                         stream_ptr(x)))
        return gx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        x, B, d_row = ctx.saved_tensors
        n, width = d_row.shape
        gg = gg.contiguous()
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dd = torch.empty((n, width), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None
        if d_x is not None or dd is not None:
            check(lib().hm_fourier_bwd_input_bwd(dptr(x), n, dptr(B), B.shape[1], dptr(d_row), _ld(d_row), dptr(gg),
                                                 dptr(d_x), dptr(dd), width, width, stream_ptr(x)))
        return d_x, None, dd


def embed_row_input_grad(x, table, B, desc, collector=None):
    """[N,E] embedding row in ONE kernel, differentiable w.r.t. the table (any order) and w.r.t. x (twice) -
    reference frac mode."""
    x = _prep_x(x)
    row = _HashFeatures.apply(x.detach(), table, B, desc, 0, collector)
    return _AttachInputGrad.apply(row, x, B)


def hash_features(x, table, desc, frac_mode=0, collector=None):
    """[N,L*F] hash features with autograd (table grads of any order)."""
    return _HashFeatures.apply(_prep_x(x), table, None, desc, frac_mode, collector)


def encode_table_grad(x, table, B, desc, frac_mode=0, collector=None):
    """Full [N,E] embedding in ONE kernel, differentiable w.r.t. the table only (x is a constant)."""
    return _HashFeatures.apply(_prep_x(x).detach(), table, B, desc, frac_mode, collector)


# =========================================================================================
# exact-fp32 MFMA GEMM (csrc/hm_gemm.hip) as an any-order differentiable torch op
# =========================================================================================
def gemm(a, b, bias=None, trans_a=False, trans_b=False, out=None, accumulate=False):
    """C = op(a) @ op(b) (+ bias) on the HIP kernel; no autograd."""
    require_gpu(a, b, bias)
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise TypeError("hashmod gemm: fp32 only")
    if a.stride(-1) != 1:
        a = a.contiguous()
    if b.stride(-1) != 1:
        b = b.contiguous()
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != Kb:
        raise ValueError(f"hashmod gemm: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if bias is not None:
        bias = bias.contiguous()
    check(lib().hm_gemm_f32(int(trans_a), int(trans_b), M, N, K, dptr(a), _ld(a), dptr(b), _ld(b), dptr(bias),
                            dptr(out), _ld(out), int(accumulate), stream_ptr(a)))
    return out


def gemm_group_tn(problems):
    """C += A^T @ B for every (A [K, M], B [K, N], C [M, N]) of `problems` in ONE launch (hm_gemm_f32_group_tn): the
    weight gradients of one backward pass.  Row-major fp32 views with arbitrary row strides; C is accumulated into."""
    if not problems:
        return
    items = (_lib.GemmGroupItem * len(problems))()
    keep = []
    for i, (a, b, c) in enumerate(problems):
        require_gpu(a, b, c)
        a, b = _rowmajor(a), _rowmajor(b)
        if c.stride(-1) != 1 or a.dtype != torch.float32 or b.dtype != torch.float32 or c.dtype != torch.float32:
            raise ValueError("hashmod gemm_group_tn: fp32 row-major operands expected")
        K, M = a.shape
        Kb, N = b.shape
        if K != Kb or tuple(c.shape) != (M, N):
            raise ValueError(f"hashmod gemm_group_tn: shapes {tuple(a.shape)}^T @ {tuple(b.shape)} -> {tuple(c.shape)}")
        keep += [a, b]
        items[i] = _lib.GemmGroupItem(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, _ld(a), _ld(b), _ld(c))
    check(lib().hm_gemm_f32_group_tn(C.cast(items, C.c_void_p), len(problems), stream_ptr(problems[0][0])))


EPI_SOFTPLUS, EPI_S1MUL, EPI_ADJOINT, EPI_RELU, EPI_RELUMASK = 1, 2, 3, 4, 5


def _rowmajor(t):
    return t if t.stride(-1) == 1 else t.contiguous()


def _ld(t):
    """row stride in elements; torch reports arbitrary strides for size-1 dimensions"""
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1], 1)


def gemm_ep(a, b, bias, trans_a, trans_b, mode, beta, thr, scale=1.0, z=None, g=None, nz=None, want_c=True,
            want_out3=False, out1=None, out3=None):
    """GEMM with a fused Softplus epilogue (hm_gemm_f32_ep; include/hashmod.h).  v = (op(a) @ op(b) + bias) * scale.
      EPI_SOFTPLUS -> (v, softplus(v))
      EPI_S1MUL    -> (v or None, v[:, :nz] * s1(z) (+ g))
      EPI_ADJOINT  -> (v * s1(z), v * g * s2(z), g * s1(z) or None)
      EPI_RELU     -> (v or None, max(v, 0));   EPI_RELUMASK -> (v or None, where(z > 0, v, 0)[:, :nz] (+ g))
    Row-major 2-D operands with arbitrary row strides (views of wider tensors are fine); no autograd.
    out1 / out3: optional caller-owned destinations (row slices of larger buffers) for the first / third output."""
    require_gpu(a, b, bias, z, g)
    a, b = _rowmajor(a), _rowmajor(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != Kb:
        raise ValueError(f"hashmod gemm: inner dimensions differ ({K} vs {Kb})")
    dev = a.device
    new = lambda cols: torch.empty((M, cols), dtype=torch.float32, device=dev)  # noqa: E731
    ep = _lib.GemmEpilogue()
    ep.mode, ep.scale, ep.beta, ep.threshold = mode, float(scale), float(beta), float(thr)
    c = new(N) if (want_c or mode == EPI_SOFTPLUS) else None
    masked = mode in (EPI_S1MUL, EPI_RELUMASK)
    outs = ()
    def dest(buf, cols):
        if buf is None:
            return new(cols)
        if buf.shape != (M, cols) or buf.stride(1) != 1 or buf.dtype != torch.float32:
            raise ValueError("hashmod gemm_ep: bad output buffer")
        return buf

    if mode in (EPI_SOFTPLUS, EPI_RELU):
        o1 = dest(out1, N)
        outs = (c, o1)
    else:
        z = _rowmajor(z)
        g = _rowmajor(g) if g is not None else None
        ncol = (z.shape[1] if nz is None else nz) if masked else N
        if z.shape[0] != M or z.shape[1] < ncol or (g is not None and (g.shape[0] != M or g.shape[1] < ncol)):
            raise ValueError("hashmod gemm_ep: epilogue operand shape")
        ep.nz = ncol
        ep.z, ep.ldz = z.data_ptr(), z.stride(0)
        if g is not None:
            ep.g, ep.ldg = g.data_ptr(), g.stride(0)
        o1 = dest(out1, ncol)
        if masked:
            outs = (c, o1)
        else:
            o2 = new(N)
            o3 = dest(out3, N) if (want_out3 or out3 is not None) else None
            ep.out2, ep.ld2 = o2.data_ptr(), o2.stride(0)
            if o3 is not None:
                ep.out3, ep.ld3 = o3.data_ptr(), o3.stride(0)
            outs = (o1, o2, o3)
    ep.out1, ep.ld1 = o1.data_ptr(), o1.stride(0)
    if bias is not None:
        bias = bias.contiguous()
    check(lib().hm_gemm_f32_ep(int(trans_a), int(trans_b), M, N, K, dptr(a), _ld(a), dptr(b), _ld(b), dptr(bias),
                               dptr(c), _ld(c) if c is not None else N, C.byref(ep), stream_ptr(a)))
    return outs


_INPUT_GRAD_ONLY = [False]


class input_grad_only:
    """`with ops.input_grad_only(): torch.autograd.grad(e, x, ..., create_graph=True)` - the caller asks for the gradient
    w.r.t. the POINTS only.  A custom Function cannot see which of its inputs were asked for (needs_input_grad only says
    which require grad), so without this hint every Linear of an embedder also forms its weight / bias gradient in that
    pass (a split-K GEMM, its zeroing launch and a column sum each) just to have it thrown away."""

    def __enter__(self):
        self.prev = _INPUT_GRAD_ONLY[0]
        _INPUT_GRAD_ONLY[0] = True

    def __exit__(self, *exc):
        _INPUT_GRAD_ONLY[0] = self.prev


class _MatMul(torch.autograd.Function):
    """C = op(A) @ op(B) + bias.  backward is expressed with the same op, so autograd can
    differentiate it again (ImplicitNetwork.gradient uses create_graph=True)."""

    @staticmethod
    def forward(ctx, a, b, bias, trans_a, trans_b):
        ctx.ta, ctx.tb = trans_a, trans_b
        ctx.save_for_backward(a, b)
        ctx.has_bias = bias is not None
        ctx.is_param = tuple(isinstance(t, torch.nn.Parameter) for t in (a, b, bias))
        return gemm(a, b, bias, trans_a, trans_b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        da = db = dbias = None
        # (parameters are leaves: they cannot lie on a path to the points)
        skip = ctx.is_param if (_INPUT_GRAD_ONLY[0] and torch.is_grad_enabled()) else (False, False, False)
        if ctx.needs_input_grad[0] and not skip[0]:
            # dA' = dC B'^T ; stored layout follows ta
            da = matmul(b, dc, None, tb, True) if ta else matmul(dc, b, None, False, not tb)
        if ctx.needs_input_grad[1] and not skip[1]:
            db = matmul(dc, a, None, True, ta) if tb else matmul(a, dc, None, not ta, False)
        if ctx.has_bias and ctx.needs_input_grad[2] and not skip[2]:
            dbias = colsum(dc)
        return da, db, dbias, None, None


def matmul(a, b, bias=None, trans_a=False, trans_b=False):
    return _MatMul.apply(a, b, bias, trans_a, trans_b)


def linear(x, weight, bias=None):
    """x [N,in] @ weight[out,in]^T + bias - nn.Linear on the HIP GEMM, differentiable to any order."""
    return _MatMul.apply(x, weight, bias, False, True)


# =========================================================================================
# fused no-grad SDF forward (csrc/hm_sdf.hip)
# =========================================================================================
def pack_mlp_layer(W, segs, kblock=8):
    """Packed MFMA operand image of one folded layer (layout contract: include/hashmod.h).

    W [out, sum(real segment widths)];  segs = [(src, real_width), ...] with src 1 = embedding,
    0 = previous layer output.  kblock 8: image for the 64-point kernel (32-row tiles, octets);
    kblock 16: image for the 16-point kernel (16-row tiles, 16-wide k blocks).
    Returns (w_packed, n_tiles, seg_blocks, seg_src)."""
    out_dim = W.shape[0]
    n_tiles = (out_dim + 31) // 32
    cols, c0, octs, srcs = [], 0, [], []
    for src, width in segs:
        pad = (-width) % kblock
        blk = W[:, c0:c0 + width]
        if pad:
            blk = torch.nn.functional.pad(blk, (0, pad))
        cols.append(blk)
        octs.append((width + pad) // kblock)
        srcs.append(src)
        c0 += width
    assert c0 == W.shape[1]
    Wp = torch.cat(cols, 1)
    Wp = torch.nn.functional.pad(Wp, (0, 0, 0, n_tiles * 32 - out_dim))
    G = Wp.shape[1] // kblock
    if kblock == 8:   # [u][g][h][i][s]  <- W[32u+i][8g+4h+s]
        Wp = Wp.view(n_tiles, 32, G, 2, 4).permute(0, 2, 3, 1, 4).contiguous()
    else:             # [u][t][q][i][e]  <- W[16u+i][16t+4q+e]
        Wp = Wp.view(n_tiles * 2, 16, G, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
    while len(octs) < 2:
        octs.append(0)
        srcs.append(0)
    return Wp, n_tiles, octs, srcs


class PackedSdf:
    """Device-resident packed weights + the [host] hm_mlp_desc for hm_sdf_fwd.

    The image buffers are allocated once (stable pointers, so the descriptor stays valid across
    optimizer steps and inside captured graphs); update() re-packs them with one kernel per layer."""

    def __init__(self, weights, biases, E, skip_in, beta, with_bf16=False, split=None):
        n = len(weights)
        self.has_bf16 = bool(with_bf16)
        self.bf16 = []
        if split not in (None, "bf16x2", "f16x2"):
            raise ValueError("hashmod PackedSdf: split must be None, 'bf16x2' or 'f16x2'")
        self.split = split                  # kind of the (hi, lo) operand images for hm_sdf_fwd_split, or None
        self.split_imgs, self.split_scales = [], []
        self.desc = _lib.MlpDesc()
        self.desc.split_kind = {None: -1, "bf16x2": 0, "f16x2": 1}[split]
        self.desc.n_layers = n
        self.keep, self.segs, self.bufs = [], [], []
        dev = weights[0].device
        prev_out = None
        for l in range(n):
            out_dim = weights[l].shape[0]
            if l == 0:
                w0, w1, srcs = E, 0, (1, 0)
            elif l in skip_in:
                w0, w1, srcs = prev_out, E, (0, 1)
            else:
                w0, w1, srcs = prev_out, 0, (0, 0)
            assert weights[l].shape[1] == w0 + w1
            n_tiles = (out_dim + 31) // 32
            oct0, oct1 = (w0 + 7) // 8, (w1 + 7) // 8
            b0, b1 = (w0 + 15) // 16, (w1 + 15) // 16
            img8 = torch.empty(n_tiles * (oct0 + oct1) * 256, dtype=torch.float32, device=dev)
            img16 = torch.empty(2 * n_tiles * (b0 + b1) * 256, dtype=torch.float32, device=dev)
            bpad = torch.empty(n_tiles * 32, dtype=torch.float32, device=dev)
            self.bufs.append((img8, img16, bpad))
            self.segs.append((out_dim, w0, w1))
            ly = self.desc.layer[l]
            ly.w_packed, ly.bias, ly.w_packed_m16 = img8.data_ptr(), bpad.data_ptr(), img16.data_ptr()
            if with_bf16:    # operand image of hm_sdf_fwd_bf16 (2-byte elements; torch.bfloat16 as the container)
                ib = torch.empty(n_tiles * (b0 + b1) * 512, dtype=torch.bfloat16, device=dev)
                self.bf16.append(ib)
                ly.w_packed_bf16 = ib.data_ptr()
            else:
                ly.w_packed_bf16 = None
            if split is not None:  # [tile][16-block][hi | lo][64 lanes][8] 2-byte elements (int16 as the container)
                isp = torch.empty(n_tiles * (b0 + b1) * 1024, dtype=torch.int16, device=dev)
                self.split_imgs.append(isp)
                ly.w_packed_split = isp.data_ptr()
                # the skip layer consumes cat[x, emb]/sqrt(2): x is divided by the previous layer's epilogue, the
                # embedding segment carries the factor in its weights
                self.split_scales.append((1.0, 1.0 / math.sqrt(2.0)) if l in skip_in else (1.0, 1.0))
            else:
                ly.w_packed_split = None
            ly.out_dim, ly.n_tiles = out_dim, n_tiles
            ly.seg_octets[0], ly.seg_octets[1] = oct0, oct1
            ly.seg_blocks16[0], ly.seg_blocks16[1] = b0, b1
            ly.seg_src[0], ly.seg_src[1] = srcs
            ly.activation = 1 if l < n - 1 else 0
            ly.post_div_sqrt2 = 1 if (l + 1) in skip_in else 0
            prev_out = out_dim
        self.out_dim = prev_out
        self.update(weights, biases, beta)

    def update(self, weights, biases, beta):
        self.desc.beta = float(beta)
        items = (_lib.PackItem * len(self.bufs))()
        self.keep = []
        for l, (img8, img16, bpad) in enumerate(self.bufs):
            out_dim, w0, w1 = self.segs[l]
            W = weights[l].detach()
            b = biases[l].detach()
            if W.dtype != torch.float32 or W.stride(-1) != 1:
                W = W.float().contiguous()
            if b.dtype != torch.float32 or not b.is_contiguous():
                b = b.float().contiguous()
            require_gpu(W, b)
            self.keep += [W, b]
            items[l] = _lib.PackItem(W.data_ptr(), b.data_ptr(), img8.data_ptr(), img16.data_ptr(), bpad.data_ptr(),
                                     W.stride(0), out_dim, w0, w1, 0)
            if self.has_bf16:
                check(lib().hm_pack_mlp_layer_bf16(dptr(W), W.stride(0), out_dim, w0, w1, dptr(self.bf16[l]),
                                                   stream_ptr(W)))
            if self.split is not None:
                s0, s1 = self.split_scales[l]
                check(lib().hm_pack_mlp_layer_split(dptr(W), W.stride(0), out_dim, w0, w1, s0, s1, self.desc.split_kind,
                                                    dptr(self.split_imgs[l]), stream_ptr(W)))
        # all fp32 operand images (8-k, 16-k) and padded biases in ONE launch
        check(lib().hm_pack_mlp_layers(C.cast(items, C.c_void_p), len(self.bufs), stream_ptr(self.bufs[0][0])))


def sdf_fwd(desc, packed, x, table, B, frac_mode=0, sdf_only=False, max_workgroups=0, tile_points=0, n_dev=None):
    """Fused encode + MLP + clamp.  Returns [N] (sdf_only) or [N, out_dim].
    tile_points 0 = auto (16-point tiles for small batches, 64 otherwise); n_dev = optional device
    int32 tensor holding the live point count (the call is then sync-free for device-compacted work)."""
    x = _prep_x(x)
    require_gpu(x, table, B)
    n = x.shape[0]
    cols = 1 if sdf_only else packed.out_dim
    out = torch.empty((n, cols), dtype=torch.float32, device=x.device)
    check(lib().hm_sdf_fwd(desc.handle, C.byref(packed.desc), dptr(x), n, dptr(table), dptr(B.contiguous()),
                           dptr(out), cols, cols, int(frac_mode), int(tile_points), dptr(n_dev),
                           int(max_workgroups), stream_ptr(x)))
    return out[:, 0] if sdf_only else out


def sdf_fwd_bf16(desc, packed, x, table, B, frac_mode=0, n_dev=None, run_min=0):
    """sdf-only values [N] from the bf16 variant of the fused kernel (hm_sdf_fwd_bf16; coarse-search precision)."""
    x = _prep_x(x)
    require_gpu(x, table, B)
    if not packed.has_bf16:
        raise ValueError("hashmod sdf_fwd_bf16: the packed weights carry no bf16 image")
    n = x.shape[0]
    out = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    check(lib().hm_sdf_fwd_bf16(desc.handle, C.byref(packed.desc), dptr(x), n, dptr(table), dptr(B.contiguous()),
                                dptr(out), 1, int(frac_mode), dptr(n_dev), int(run_min), stream_ptr(x)))
    return out[:, 0]


def sdf_fwd_split(desc, packed, x, table, B, frac_mode=0, n_dev=None, run_min=0):
    """sdf-only values [N] from the split-operand kernel (hm_sdf_fwd_split; kind = packed.split)."""
    x = _prep_x(x)
    require_gpu(x, table, B)
    if packed.split is None:
        raise ValueError("hashmod sdf_fwd_split: the packed weights carry no split image")
    n = x.shape[0]
    out = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    check(lib().hm_sdf_fwd_split(desc.handle, C.byref(packed.desc), dptr(x), n, dptr(table), dptr(B.contiguous()),
                                 dptr(out), 1, int(frac_mode), dptr(n_dev), int(run_min), stream_ptr(x)))
    return out[:, 0]


def sdf_fwd_emb_split(packed, emb, n_dev=None, run_min=0):
    """the same on precomputed embedding rows (hm_sdf_fwd_emb_split)"""
    require_gpu(emb)
    if packed.split is None:
        raise ValueError("hashmod sdf_fwd_emb_split: the packed weights carry no split image")
    if emb.stride(-1) != 1:
        emb = emb.contiguous()
    n, width = emb.shape
    out = torch.empty((n, 1), dtype=torch.float32, device=emb.device)
    check(lib().hm_sdf_fwd_emb_split(C.byref(packed.desc), dptr(emb), emb.stride(0), width, n, dptr(out), 1,
                                     dptr(n_dev), int(run_min), stream_ptr(emb)))
    return out[:, 0]


def sdf_fwd_emb_bf16(packed, emb, n_dev=None, run_min=0):
    require_gpu(emb)
    if emb.stride(-1) != 1:
        emb = emb.contiguous()
    n, width = emb.shape
    out = torch.empty((n, 1), dtype=torch.float32, device=emb.device)
    check(lib().hm_sdf_fwd_emb_bf16(C.byref(packed.desc), dptr(emb), emb.stride(0), width, n, dptr(out), 1,
                                    dptr(n_dev), int(run_min), stream_ptr(emb)))
    return out[:, 0]


def sdf_fwd_emb(packed, emb, sdf_only=False, tile_points=0, n_dev=None, max_workgroups=0):
    """The fused MLP + clamp on PRECOMPUTED embedding rows (hm_sdf_fwd_emb): SDF networks whose embedder is not
    the plain hash grid (FourierFilterBanks via nffb_fwd)."""
    require_gpu(emb)
    if emb.stride(-1) != 1:
        emb = emb.contiguous()
    n, width = emb.shape
    cols = 1 if sdf_only else packed.out_dim
    out = torch.empty((n, cols), dtype=torch.float32, device=emb.device)
    check(lib().hm_sdf_fwd_emb(C.byref(packed.desc), dptr(emb), emb.stride(0), width, n, dptr(out), cols, cols,
                               int(tile_points), dptr(n_dev), int(max_workgroups), stream_ptr(emb)))
    return out[:, 0] if sdf_only else out


# =========================================================================================
# fused Fourier-filter-bank embedder forward (csrc/hm_nffb.hip)
# =========================================================================================
class NffbPacked:
    """[host] hm_nffb_desc over the live parameters of a FourierFilterBanks module (no copies: the kernel reads the
    nn.Linear weights in place, so optimizer updates are seen without re-packing)."""

    def __init__(self, mod):
        self.desc = _lib.NffbDesc()
        self.refresh(mod)

    def refresh(self, mod):
        d = self.desc
        L = mod.n_levels
        d.n_levels, d.bound, d.w0, d.style_eps = int(L), float(mod.bound), float(mod.sin_w0), 1e-5
        keep = []
        for l in range(L - 1):
            lin = getattr(mod, "ff_lin" + str(l))
            w, b = lin.weight.detach(), lin.bias.detach()
            require_gpu(w, b)
            if not (w.is_contiguous() and b.is_contiguous() and w.dtype == torch.float32):
                raise ValueError("hashmod nffb: contiguous fp32 parameters expected")
            d.trunk_w[l], d.trunk_b[l] = w.data_ptr(), b.data_ptr()
            keep += [w, b]
        d.out_w, d.out_b = mod.out_layer.weight.data_ptr(), mod.out_layer.bias.data_ptr()
        if mod.modulationApplied:
            lt = mod.StyleAttentionBlock.linear_transform
            d.style_w, d.style_b = lt.weight.data_ptr(), lt.bias.data_ptr()
            d.style_eps = float(mod.StyleAttentionBlock.eps)
        else:
            d.style_w, d.style_b = None, None
        self.key = tuple(t.data_ptr() for t in keep)


def nffb_packed(mod):
    """the module's hm_nffb_desc, pointers refreshed (parameters may have been moved / re-bound since the last call)"""
    pk = mod.__dict__.get("_nffb_packed")
    if pk is None:
        pk = mod.__dict__["_nffb_packed"] = NffbPacked(mod)
    else:
        pk.refresh(mod)
    return pk


def nffb_fwd(mod, x, n_dev=None, out=None):
    """[N, 3 + 8 + 8L] embedding of a FourierFilterBanks module in one kernel (no autograd)."""
    x = _prep_x(x)
    grid = mod.grid_enc
    require_gpu(x, grid.table)
    n = x.shape[0]
    width = mod.embeddings_dim
    if out is None:
        out = torch.empty((n, width), dtype=torch.float32, device=x.device)
    pk = nffb_packed(mod)
    check(lib().hm_nffb_fwd(grid.desc.handle, C.byref(pk.desc), dptr(x), n, dptr(grid.table.detach()),
                            dptr(grid.freq_encoding.B), dptr(out), out.stride(0), FRAC_MODES[grid.frac_mode],
                            dptr(n_dev), stream_ptr(x)))
    return out


# =========================================================================================
# sync-free ray tracer (csrc/hm_trace.hip)
# =========================================================================================
def trace_workspace_bytes(n_rays, cfg, nffb_levels=0):
    if nffb_levels:
        return check(lib().hm_trace_workspace_bytes_nffb(int(n_rays), C.byref(cfg), int(nffb_levels)))
    return check(lib().hm_trace_workspace_bytes(int(n_rays), C.byref(cfg)))


def camera_rays(uv, pose, intrinsics, radius):
    """(ray_dirs [B,N,3], cam_loc [B,3], t_sphere [B,N,2], hit [B,N] bool) of fixed 4x4 cameras in one launch
    (hm_camera_rays: rend_util.get_camera_params + get_sphere_intersection; no gradient)."""
    require_gpu(uv, pose, intrinsics)
    Bn, N = int(uv.shape[0]), int(uv.shape[1])
    if tuple(pose.shape[1:]) != (4, 4) or tuple(intrinsics.shape[1:]) != (4, 4):
        raise ValueError("hashmod camera_rays: pose and intrinsics must be [B,4,4]")
    uv_c, pose_c, k_c = uv.detach().float().contiguous(), pose.detach().float().contiguous(), intrinsics.detach().float().contiguous()
    dev = uv.device
    dirs = torch.empty((Bn, N, 3), dtype=torch.float32, device=dev)
    cam = torch.empty((Bn, 3), dtype=torch.float32, device=dev)
    t = torch.empty((Bn, N, 2), dtype=torch.float32, device=dev)
    hit = torch.empty((Bn, N), dtype=torch.uint8, device=dev)
    check(lib().hm_camera_rays(dptr(uv_c), dptr(pose_c), dptr(k_c), Bn, N, float(radius), dptr(dirs), dptr(cam), dptr(t),
                               dptr(hit), stream_ptr(uv_c)))
    return dirs, cam, t, hit.bool()


def trace_forward(desc, packed, table, B, frac_mode, tile_points, cfg, cam_loc, ray_dirs, object_mask, t_sphere,
                  hit_mask, rays_per_image, sampler_fracs, steps_u, workspace, stats=None, nffb=None):
    """Enqueues the whole intersection search (no host sync).  Returns (points, net_mask_u8, dists).
    nffb: NffbPacked of a FourierFilterBanks embedder (desc / table / B are then its hash grid's)."""
    require_gpu(cam_loc, ray_dirs, object_mask, t_sphere, hit_mask, sampler_fracs, workspace)
    n = ray_dirs.shape[0]
    dev = ray_dirs.device
    pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
    mask = torch.empty((n,), dtype=torch.uint8, device=dev)
    dists = torch.empty((n,), dtype=torch.float32, device=dev)
    tail = (dptr(table), dptr(B), int(frac_mode), int(tile_points), C.byref(cfg), dptr(cam_loc), dptr(ray_dirs),
            dptr(object_mask), dptr(t_sphere), dptr(hit_mask), n, int(rays_per_image), dptr(sampler_fracs),
            dptr(steps_u), dptr(pts), dptr(mask), dptr(dists), dptr(workspace), workspace.numel(), dptr(stats),
            stream_ptr(ray_dirs))
    if nffb is not None:
        check(lib().hm_trace_forward_nffb(desc.handle, C.byref(nffb.desc), C.byref(packed.desc), *tail))
    else:
        check(lib().hm_trace_forward(desc.handle, C.byref(packed.desc), *tail))
    return pts, mask, dists


# =========================================================================================
# fused elementwise passes (csrc/hm_elem.hip)
# =========================================================================================
class _ColSum(torch.autograd.Function):
    """sum over rows (bias gradient) in one pass; its backward is a broadcast."""

    @staticmethod
    def forward(ctx, x):
        ctx.rows = x.shape[0]
        if x.stride(-1) != 1:
            x = x.contiguous()
        out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
        check(lib().hm_colsum(dptr(x), x.shape[0], x.shape[1], max(x.stride(0), 1), dptr(out), stream_ptr(x)))
        return out

    @staticmethod
    def backward(ctx, g):
        return g.unsqueeze(0).expand(ctx.rows, -1)


def colsum(x):
    require_gpu(x)
    return _ColSum.apply(x)


def sdf_head_fwd(zl, beta_rho):
    """(out, sdf, c, denom) of the soft SDF clamp (hm_sdf_head; no autograd)."""
    require_gpu(zl)
    zl = zl.contiguous()
    n, cols = zl.shape
    out = torch.empty_like(zl)
    sdf, c, denom = (torch.empty(n, dtype=torch.float32, device=zl.device) for _ in range(3))
    check(lib().hm_sdf_head(0, dptr(zl), n, cols, float(beta_rho), dptr(out), dptr(sdf), dptr(c), dptr(denom),
                            None, stream_ptr(zl)))
    return out, sdf, c, denom


def sdf_head_bwd(d_out, sdf, c, denom, cb=None):
    """z-bar of the last layer from d_out (and the gradient sweep's c-bar, if any)."""
    require_gpu(d_out, cb)
    d_out = d_out.contiguous()
    n, cols = d_out.shape
    zb = torch.empty_like(d_out)
    cbp = cb.contiguous() if cb is not None else None
    check(lib().hm_sdf_head(1, dptr(d_out), n, cols, 1.0, dptr(zb), dptr(sdf), dptr(c), dptr(denom), dptr(cbp),
                            stream_ptr(d_out)))
    return zb


def dcopy_(dst, src):
    """dst.copy_(src) for 2-D (or 1-D) fp32 row-major tensors as a KERNEL launch (hm_copy2d_f32).  torch copies
    contiguous tensors with hipMemcpyAsync, which a graph capture records as a MEMCPY node."""
    require_gpu(dst, src)
    if dst.shape != src.shape or dst.dtype != torch.float32 or src.dtype != torch.float32:
        raise ValueError("hashmod dcopy_: shape / dtype mismatch")
    d2 = dst if dst.dim() == 2 else dst.reshape(1, -1) if dst.is_contiguous() else None
    s2 = src if src.dim() == 2 else src.reshape(1, -1) if src.is_contiguous() else None
    if d2 is None or s2 is None or d2.stride(-1) != 1 or s2.stride(-1) != 1:
        raise ValueError("hashmod dcopy_: row-major 2-D (or contiguous) tensors only")
    rows, cols = d2.shape
    check(lib().hm_copy2d_f32(dptr(d2), _ld(d2), dptr(s2), _ld(s2), rows, cols, stream_ptr(dst)))
    return dst


class _TakeBlock(torch.autograd.Function):
    """y = x[r0:r0+nr, c0:c0+nc] as a fresh contiguous tensor.  Plain slicing is a view whose autograd backward
    (SliceBackward: zeros + narrow().copy_()) copies contiguous rows with hipMemcpyAsync, i.e. a MEMCPY node in a
    captured graph; here both directions are kernel copies (hm_copy2d_f32).  Differentiable to any order."""

    @staticmethod
    def forward(ctx, x, r0, nr, c0, nc):
        ctx.box = (x.shape, r0, nr, c0, nc)
        y = torch.empty((nr, nc), dtype=torch.float32, device=x.device)
        return dcopy_(y, x[r0:r0 + nr, c0:c0 + nc])

    @staticmethod
    def backward(ctx, dy):
        shape, r0, nr, c0, nc = ctx.box
        return _PutBlock.apply(dy, shape, r0, nr, c0, nc), None, None, None, None


class _PutBlock(torch.autograd.Function):
    """zeros(shape) with dy written at [r0:r0+nr, c0:c0+nc] (the adjoint of _TakeBlock)."""

    @staticmethod
    def forward(ctx, dy, shape, r0, nr, c0, nc):
        ctx.box = (r0, nr, c0, nc)
        dx = torch.zeros(shape, dtype=torch.float32, device=dy.device)      # fill kernel
        dcopy_(dx[r0:r0 + nr, c0:c0 + nc], dy if dy.stride(-1) == 1 else dy.contiguous())
        return dx

    @staticmethod
    def backward(ctx, ddx):
        r0, nr, c0, nc = ctx.box
        return _TakeBlock.apply(ddx, r0, nr, c0, nc), None, None, None, None, None


def take_block(x, r0, nr, c0=0, nc=None):
    """x[r0:r0+nr, c0:c0+nc] of a 2-D fp32 tensor as a new tensor, with kernel copies in forward and backward."""
    require_gpu(x)
    if x.dim() != 2 or x.dtype != torch.float32:
        raise ValueError("hashmod take_block: 2-D fp32 tensor expected")
    if x.stride(-1) != 1:
        x = x.contiguous()
    nc = x.shape[1] - c0 if nc is None else nc
    return _TakeBlock.apply(x, int(r0), int(nr), int(c0), int(nc))


def cat_rows_(dst, parts, dim):
    """torch.cat(parts, dim, out=dst) for 2-D fp32 tensors through dcopy_ (no MEMCPY graph nodes)."""
    o = 0
    for t in parts:
        n = t.shape[dim]
        dcopy_(dst.narrow(dim, o, n), t if t.stride(-1) == 1 else t.contiguous())
        o += n
    return dst


def colsum_into(x, out):
    """out += column sums of x (no autograd; `out` zeroed by the caller)."""
    require_gpu(x, out)
    x = _rowmajor(x)
    check(lib().hm_colsum_acc(dptr(x), x.shape[0], x.shape[1], _ld(x), dptr(out), stream_ptr(x)))
    return out


def colsum_into_multi(pairs):
    """out += column sums of x for every (x, out) of `pairs` in ONE launch (hm_colsum_acc_multi)."""
    if not pairs:
        return
    items = (_lib.ColsumItem * len(pairs))()
    keep = []
    for i, (x, out) in enumerate(pairs):
        require_gpu(x, out)
        x = _rowmajor(x)
        keep.append(x)
        items[i] = _lib.ColsumItem(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _ld(x))
    check(lib().hm_colsum_acc_multi(C.cast(items, C.c_void_p), len(pairs), stream_ptr(pairs[0][0])))


def _softplus_call(order, z, gy, gg, beta, thr):
    n = z.numel()
    out0 = torch.empty_like(z)
    out1 = torch.empty_like(z) if order == 2 else None
    check(lib().hm_softplus(order, dptr(z), dptr(gy), dptr(gg), dptr(out0), dptr(out1), n, float(beta), float(thr),
                            stream_ptr(z)))
    return out0, out1


class _Softplus(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, beta, thr):
        z = z.contiguous()
        ctx.beta, ctx.thr = beta, thr
        ctx.save_for_backward(z)
        return _softplus_call(0, z, None, None, beta, thr)[0]

    @staticmethod
    def backward(ctx, gy):
        (z,) = ctx.saved_tensors
        return _SoftplusBwd.apply(z, gy, ctx.beta, ctx.thr), None, None


class _SoftplusBwd(torch.autograd.Function):
    """gz = gy * s1(z); differentiable once more (d/dgy and d/dz in one fused pass)."""

    @staticmethod
    def forward(ctx, z, gy, beta, thr):
        gy = gy.contiguous()
        ctx.beta, ctx.thr = beta, thr
        ctx.save_for_backward(z, gy)
        return _softplus_call(1, z, gy, None, beta, thr)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        z, gy = ctx.saved_tensors
        d_gy, d_z = _softplus_call(2, z, gy, gg.contiguous(), ctx.beta, ctx.thr)
        return d_z, d_gy, None, None


def _sine_call(order, x, gy, gg, w0):
    out0 = torch.empty_like(x)
    out1 = torch.empty_like(x) if order == 2 else None
    check(lib().hm_sine(order, dptr(x), dptr(gy), dptr(gg), dptr(out0), dptr(out1), x.numel(), float(w0), stream_ptr(x)))
    return out0, out1


class _Sine(torch.autograd.Function):
    """sin(w0 x) (SIREN activation) with one-kernel backward and double backward (csrc/hm_elem.hip)."""

    @staticmethod
    def forward(ctx, x, w0):
        x = x.contiguous()
        ctx.w0 = w0
        ctx.save_for_backward(x)
        return _sine_call(0, x, None, None, w0)[0]

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _SineBwd.apply(x, gy, ctx.w0), None


class _SineBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gy, w0):
        gy = gy.contiguous()
        ctx.w0 = w0
        ctx.save_for_backward(x, gy)
        return _sine_call(1, x, gy, None, w0)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        x, gy = ctx.saved_tensors
        d_gy, d_x = _sine_call(2, x, gy, gg.contiguous(), ctx.w0)
        return d_x, d_gy, None


def sine(x, w0):
    """sin(w0 * x), differentiable twice (third order is not provided)."""
    require_gpu(x)
    return _Sine.apply(x, float(w0))


class _WeightNormFold(torch.autograd.Function):
    """W_l = g_l * v_l / ||v_l||_row for a LIST of weight-normed layers: one launch forward, one backward
    (torch._weight_norm is one launch per layer and pass)."""

    @staticmethod
    def forward(ctx, *vg):
        L = len(vg) // 2
        vs = [t.contiguous() for t in vg[:L]]
        gs = [t.contiguous() for t in vg[L:]]
        ws = [torch.empty_like(v) for v in vs]
        norms = [torch.empty(v.shape[0], dtype=torch.float32, device=v.device) for v in vs]
        tab = (_lib.WnLayer * L)()
        for i in range(L):
            tab[i].v, tab[i].g, tab[i].w, tab[i].norm = vs[i].data_ptr(), gs[i].data_ptr(), ws[i].data_ptr(), norms[i].data_ptr()
            tab[i].rows, tab[i].cols = vs[i].shape[0], vs[i].shape[1]
        check(lib().hm_weight_norm_multi(0, L, C.cast(tab, C.c_void_p), stream_ptr(vs[0])))
        ctx.L = L
        ctx.save_for_backward(*vs, *gs, *norms)
        return tuple(ws)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gws):
        L = ctx.L
        sv = ctx.saved_tensors
        vs, gs, norms = sv[:L], sv[L:2 * L], sv[2 * L:]
        idx = [i for i in range(L) if gws[i] is not None]
        gv = [None] * L
        gg = [None] * L
        if idx:
            keep = []
            tab = (_lib.WnLayer * len(idx))()
            for k, i in enumerate(idx):
                gw = gws[i].contiguous()
                keep.append(gw)
                gv[i] = torch.empty_like(vs[i])
                gg[i] = torch.empty_like(gs[i])
                tab[k].v, tab[k].g, tab[k].norm = vs[i].data_ptr(), gs[i].data_ptr(), norms[i].data_ptr()
                tab[k].grad_w, tab[k].grad_v, tab[k].grad_g = gw.data_ptr(), gv[i].data_ptr(), gg[i].data_ptr()
                tab[k].rows, tab[k].cols = vs[i].shape[0], vs[i].shape[1]
            check(lib().hm_weight_norm_multi(1, len(idx), C.cast(tab, C.c_void_p), stream_ptr(vs[0])))
        return (*gv, *gg)


def weight_norm_fold(vs, gs):
    """[g * v / ||v||_row for (v, g) in zip(vs, gs)] (g of shape [rows, 1] or [rows]) - differentiable once."""
    require_gpu(*vs)
    return list(_WeightNormFold.apply(*vs, *[g.reshape(-1) for g in gs]))


def _rownorm_call(order, y, g, gg, eps):
    out0 = torch.empty_like(y)
    out1 = torch.empty_like(y) if order == 2 else None
    check(lib().hm_rownorm(order, dptr(y), dptr(g), dptr(gg), dptr(out0), dptr(out1), y.shape[0], y.shape[1], float(eps),
                           stream_ptr(y)))
    return out0, out1


class _RowNorm(torch.autograd.Function):
    """(y - mean) / sqrt(var + eps) per row with one-kernel backward and double backward (csrc/hm_elem.hip)."""

    @staticmethod
    def forward(ctx, y, eps):
        y = y.contiguous()
        ctx.eps = eps
        ctx.save_for_backward(y)
        return _rownorm_call(0, y, None, None, eps)[0]

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _RowNormBwd.apply(y, g, ctx.eps), None


class _RowNormBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, g, eps):
        g = g.contiguous()
        ctx.eps = eps
        ctx.save_for_backward(y, g)
        return _rownorm_call(1, y, g, None, eps)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        y, g = ctx.saved_tensors
        d_g, d_y = _rownorm_call(2, y, g, gg.contiguous(), ctx.eps)
        return d_y, d_g, None


def rownorm(y, eps=1e-5):
    """per-row normalisation of a [N, W] tensor (biased variance), differentiable twice."""
    require_gpu(y)
    if y.dim() != 2 or y.dtype != torch.float32:
        raise ValueError("hashmod rownorm: [N, W] fp32 input")
    return _RowNorm.apply(y, float(eps))


def _posenc_call(order, freqs, c, g, gg):
    n, D = c.shape
    W = 2 * D + 2 * len(freqs) * D
    fa = (C.c_float * len(freqs))(*freqs)
    out0 = torch.empty((n, D if order == 1 else W), dtype=torch.float32, device=c.device)
    out1 = torch.empty((n, D), dtype=torch.float32, device=c.device) if order == 2 else None
    check(lib().hm_posenc(order, C.cast(fa, C.c_void_p), len(freqs), D, dptr(c), _ld(c), dptr(g),
                          _ld(g) if g is not None else 0, dptr(gg), dptr(out0), out0.stride(0), dptr(out1), n,
                          stream_ptr(c)))
    return out0, out1


class _PosEnc(torch.autograd.Function):
    """[c | c | sin(f0 c) | cos(f0 c) | ...] (NeRF positional encoding as the reference builds it with include_input)
    with one-kernel backward and double backward (csrc/hm_elem.hip)."""

    @staticmethod
    def forward(ctx, c, freqs):
        c = _rowmajor(c)
        ctx.freqs = freqs
        ctx.save_for_backward(c)
        return _posenc_call(0, freqs, c, None, None)[0]

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        return _PosEncBwd.apply(c, g, ctx.freqs), None


class _PosEncBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, g, freqs):
        g = _rowmajor(g)
        ctx.freqs = freqs
        ctx.save_for_backward(c, g)
        return _posenc_call(1, freqs, c, g, None)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        c, g = ctx.saved_tensors
        d_g, d_c = _posenc_call(2, ctx.freqs, c, g, gg.contiguous())
        return d_c, d_g, None


def posenc(c, freqs):
    """NeRF positional encoding rows (include_input form of the reference), differentiable twice."""
    require_gpu(c)
    if c.dim() != 2 or c.dtype != torch.float32:
        raise ValueError("hashmod posenc: [N, D] fp32 input")
    return _PosEnc.apply(c, tuple(float(f) for f in freqs))


def softplus(z, beta=100.0, threshold=20.0):
    """nn.Softplus(beta, threshold) with fused backward and double backward (third order is not provided)."""
    require_gpu(z)
    return _Softplus.apply(z, float(beta), float(threshold))
