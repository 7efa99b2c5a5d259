"""SDF MLP with its input-gradient as ONE autograd node with a hand-written backward.

ImplicitNetwork.gradient (reference: model/implicit_differentiable_renderer.py:116-128) asks autograd for
d sdf / d x with create_graph=True and later differentiates that gradient again (eikonal term, normals).
Done generically, every layer contributes ~30 small autograd kernels to the double backward.  Here the pair

        (e, W_0..W_{L-1}, b_0..b_{L-1})  ->  (out [N, 1+fvs],  g_e = d sdf / d e [N, E])

is a single torch.autograd.Function: the forward runs the layer GEMMs (csrc/hm_gemm.hip) and the reverse sweep
that produces g_e; the backward is the analytic reverse-over-reverse of both sweeps - per layer six GEMMs and
four fused elementwise passes (csrc/hm_elem.hip) - so nothing needs create_graph inside.  The embedding e(x)
stays outside (its Jacobian is handled by autograd: g = J_e(x)^T g_e), which is where the x-dependence of the
Fourier features and - in the reference's frac mode - the zero hash-feature gradient come from.

Notation (per layer l, N points):  a_l input (h_l, or cat[h_l, e]/sqrt2 at the skip), z_l = a_l W_l^T + b_l,
h_{l+1} = softplus(z_l), s1/s2 first/second softplus derivative at z_l;
gradient sweep: u_l = d sdf/d z_l, v_l = u_l W_l = d sdf/d a_l, u_{l-1} = v_l[:, :dh] * s1_{l-1}.
Soft clamp of the SDF column: sdf = tanh(s / (2 + rho(s))) with rho evaluated without gradient
(density_net.py:20-30), c = d sdf/d s = (1 - sdf^2)/(2 + rho), dc/ds = -2 sdf c / (2 + rho).
"""
import math

import torch

from . import ops
from .ops import (EPI_ADJOINT, EPI_RELU, EPI_RELUMASK, EPI_S1MUL, EPI_SOFTPLUS, _softplus_call, colsum_into_multi, gemm,
                  gemm_ep, gemm_group_tn)

_SQRT2 = math.sqrt(2.0)


class _SdfMlp(torch.autograd.Function):
    """The weight gradient of a hidden layer l is  u_l^T v-bar_l + z-bar_l^T a_l  (adjoint of the gradient sweep +
    backward of the forward sweep).  Both products share the point dimension, so they run as ONE GEMM over stacked
    operands [u_l; z-bar_l]^T [v-bar_l; a_l] (K = 2N): `stack[l]` is a [2N, in_l] buffer whose lower half receives
    a_l during forward (straight from the previous layer's Softplus epilogue) and whose upper half receives
    v-bar_l during backward; `ustack[l]` [2N, out_l] is filled the same way with u_l and z-bar_l."""

    @staticmethod
    def forward(ctx, e, skip_layer, beta_sp, thr_sp, beta_rho, cache_out, *params):
        L = len(params) // 2
        Ws, bs = params[:L], params[L:]
        N, E = e.shape
        new = lambda r, c: torch.empty((r, c), dtype=torch.float32, device=e.device)  # noqa: E731
        stack = [new(2 * N, Ws[l].shape[1]) for l in range(L - 1)] + [None]     # the last layer is not stacked
        lower = lambda l: stack[l][N:] if stack[l] is not None else None        # noqa: E731
        a_list, z_list = [], []
        h = None
        for l in range(L):
            if l == 0:
                a = e if stack[0] is None else ops.dcopy_(lower(0), e)   # (kernel copy: no MEMCPY graph node)
            elif l == skip_layer:
                a = torch.cat([h, e], 1, out=lower(l)).div_(_SQRT2) if stack[l] is not None \
                    else torch.cat([h, e], 1) / _SQRT2
            else:
                a = h                                   # == lower(l) when the previous epilogue wrote it there
            if l < L - 1:    # z_l and h_{l+1} = softplus(z_l) from one kernel; h goes where layer l+1 reads it
                dst = lower(l + 1) if (l + 1 != skip_layer and stack[l + 1] is not None) else None
                z, h = gemm_ep(a, Ws[l], bs[l], False, True, EPI_SOFTPLUS, beta_sp, thr_sp, out1=dst)
            else:
                z = gemm(a, Ws[l], bs[l], False, True)
            a_list.append(a)
            z_list.append(z)
        out, sdf, c, denom = ops.sdf_head_fwd(z_list[-1], beta_rho)     # soft clamp of column 0, one launch

        # reverse sweep for g_e = d sdf / d e: v_l = u_l W_l (scaled at the skip), u_{l-1} = v_l[:, :dh] * s1(z_{l-1})
        # - the product with s1 is the epilogue of the GEMM that makes v_l
        v_list = [None] * L
        v = c.unsqueeze(-1) * Ws[L - 1][0:1, :]          # u_{L-1} is c on column 0 only: rank-1, no GEMM
        v_list[L - 1] = v
        ge_skip = None
        u = _softplus_call(1, z_list[L - 2], v, None, beta_sp, thr_sp)[0] if L > 1 else None
        for l in range(L - 2, -1, -1):
            scale = 1.0 / _SQRT2 if l == skip_layer else 1.0
            if l > 0:
                dh = z_list[l - 1].shape[1]
                v, u = gemm_ep(u, Ws[l], None, False, False, EPI_S1MUL, beta_sp, thr_sp, scale=scale,
                               z=z_list[l - 1], nz=dh)
            else:
                v = gemm(u, Ws[l], None, False, False)
                dh = v.shape[1]
            if l == skip_layer:
                ge_skip = v[:, dh:]
                v = v[:, :dh]
            v_list[l] = v
        g_e = v_list[0] if ge_skip is None else v_list[0] + ge_skip

        ctx.meta = (L, skip_layer, beta_sp, thr_sp, E)
        ctx.stack = stack      # internal buffers (their lower halves are the saved a_l views)
        ctx.save_for_backward(e, sdf, c, denom, *Ws, *a_list, *z_list, *v_list)
        ctx.shared = cache_out        # (weight / bias gradient accumulators shared with a _SdfMlpRows node, see there)
        if cache_out is not None:     # lets a second node over a ROW RANGE of this batch skip its forward (_SdfMlpRows)
            cache_out.update(meta=ctx.meta, out=out, g_e=g_e, sdf=sdf, c=c, denom=denom, a_list=a_list, z_list=z_list,
                             v_list=v_list, Ws=Ws)
        return out, g_e

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_ge):
        L = ctx.meta[0]
        grads = _sdf_mlp_backward(ctx.meta, ctx.saved_tensors, ctx.stack, ctx.needs_input_grad[0],
                                  ctx.needs_input_grad[6:6 + L], ctx.needs_input_grad[6 + L:6 + 2 * L], d_out, d_ge,
                                  shared=ctx.shared, owner=True)
        return (grads[0], None, None, None, None, None, *grads[1:])


def _sdf_mlp_backward(meta, sv, stack, need_e, need_w, need_b, d_out, d_ge, shared=None, owner=True):
    """Analytic reverse-over-reverse of the forward sweep and the gradient sweep (see the module docstring).
    stack: the forward's [2N, in_l] buffers (weight gradients as ONE stacked product per layer), or None (a node that
    reuses another node's saved activations, _SdfMlpRows: u^T v-bar and z-bar^T a are two entries of the grouped launch).
    shared / owner: two nodes over the SAME weights (the evaluation and its _SdfMlpRows companion) accumulate their
    weight / bias gradients into ONE set of buffers kept in the dict `shared`: the companion (owner = False; its
    backward always runs first - its input depends on the owner's output) allocates and zeroes them and returns no
    weight gradients, the owner adds its own share and returns the sums - autograd then has nothing to add up (18
    elementwise launches per step) and the buffers are zeroed once."""
    L, skip_layer, beta_sp, thr_sp, E = meta
    e, sdf, c, denom = sv[0], sv[1], sv[2], sv[3]
    Ws = sv[4:4 + L]
    a_list = sv[4 + L:4 + 2 * L]
    z_list = sv[4 + 2 * L:4 + 3 * L]
    v_list = sv[4 + 3 * L:4 + 4 * L]
    use_stack = stack is not None
    N = e.shape[0]
    new = lambda r, c_: torch.empty((r, c_), dtype=torch.float32, device=e.device)  # noqa: E731
    # every weight / bias gradient accumulates with atomics (grouped GEMM, column sums): the buffers are zeroed
    # here by ONE multi-tensor launch instead of one zeroing launch in front of each producer
    bshape = [z.shape[1] for z in z_list]
    acc = shared.pop("wgrad_acc", None) if (shared is not None and owner) else None
    if acc is not None and acc[2] == (tuple(bool(w) for w in need_w), tuple(bool(b) for b in need_b)):
        dW, db = acc[0], acc[1]          # the companion's sums: this pass adds to them
    else:
        dW = [torch.empty_like(Ws[l]) if need_w[l] else None for l in range(L)]
        db = [torch.empty(bshape[l], dtype=torch.float32, device=e.device) if need_b[l] else None for l in range(L)]
        zero_list = [t for t in dW + db if t is not None]
        if zero_list:
            torch._foreach_zero_(zero_list)
        if acc is not None:              # (different gradient sets on the two nodes: not the static step; add up here)
            for l in range(L):
                if dW[l] is not None and acc[0][l] is not None:
                    dW[l].add_(acc[0][l])
                if db[l] is not None and acc[1][l] is not None:
                    db[l].add_(acc[1][l])
    zx = [None] * L       # extra z-bar from the adjoint of the gradient sweep
    ustack = [None] * L   # [u_l; z-bar_l] of the layers whose weight gradient is one stacked GEMM
    cb = None
    wgrad, bgrad = [], []

    # ---- adjoint of the gradient sweep (walks the layers upwards) --------------------------------------
    if d_ge is not None:
        d_ge = d_ge.contiguous()
        vb_h = d_ge                                          # v-bar of layer 0 (hidden part)
        for l in range(L):
            stacked = use_stack and l < L - 1 and bool(need_w[l])
            if l == skip_layer:
                vb = (torch.cat([vb_h, d_ge], 1, out=stack[l][:N]) if stacked else torch.cat([vb_h, d_ge], 1))
                vb = vb.div_(_SQRT2)
            elif stacked and vb_h.data_ptr() != stack[l].data_ptr():
                vb = ops.dcopy_(stack[l][:N], vb_h)          # (layer 0, or after an unstacked layer)
            else:
                vb = vb_h
            if l < L - 1:
                # u-bar_l = v-bar_l W_l^T and, on its accumulators, the adjoint of u_l = v_{l+1}[:, :dh] * s1(z_l):
                #   v-bar_{l+1} = u-bar * s1,  extra z-bar_l = u-bar * v_{l+1} * s2,  u_l itself (for W-bar_l)
                if stacked:
                    ustack[l] = new(2 * N, Ws[l].shape[0])
                nxt = l + 1
                dst = stack[nxt][:N] if (use_stack and nxt < L - 1 and nxt != skip_layer and need_w[nxt]) else None
                loose = (not stacked) and bool(need_w[l])    # u_l as a tensor of its own: u^T v-bar joins the grouped launch
                vb_h, zx[l], u_l = gemm_ep(vb, Ws[l], None, False, True, EPI_ADJOINT, beta_sp, thr_sp,
                                           z=z_list[l], g=v_list[l + 1], out1=dst,
                                           out3=ustack[l][:N] if stacked else None, want_out3=loose)
                if loose:
                    wgrad.append((u_l, vb, dW[l]))
            else:
                # last layer: u = c * onehot(0).  c-bar = u-bar[:, 0] = v-bar W[0]^T;  W-bar[0] = c^T v-bar
                # (both on the library's GEMM: torch.matmul would put a vendor GEMV and, for the row assignment,
                #  a MEMCPY node into the captured iteration)
                cb = gemm(vb, Ws[l][0:1], None, False, True)             # [N, 1]
                if need_w[l]:
                    gemm(c.reshape(N, 1), vb, None, True, False, out=dW[l][0:1], accumulate=True)

    # ---- backward of the forward sweep (walks the layers downwards) ------------------------------------
    # the weight gradients are independent of one another and of the rest of the sweep: they are collected and run as
    # ONE grouped launch at the end (hm_gemm_f32_group_tn) instead of one split-K GEMM (+ its share of launches) per layer
    zb = ops.sdf_head_bwd(d_out, sdf, c, denom, cb)
    de = None
    for l in range(L - 1, -1, -1):
        if need_w[l]:
            if ustack[l] is not None:                        # zb IS ustack[l][N:] (written by layer l+1 below)
                wgrad.append((ustack[l], stack[l], dW[l]))                                  # [u; z-bar]^T [v-bar; a]
            else:
                wgrad.append((zb, a_list[l], dW[l]))
        if need_b[l]:
            bgrad.append((zb, db[l]))                        # column sums: one launch for all layers, below
        if l > 0:
            # a-bar_l = z-bar_l W_l; z-bar_{l-1} = a-bar_l[:, :dh] * s1(z_{l-1}) (+ the adjoint sweep's share)
            dh = z_list[l - 1].shape[1]
            is_skip = l == skip_layer
            dst = ustack[l - 1][N:] if ustack[l - 1] is not None else None
            ab, zb = gemm_ep(zb, Ws[l], None, False, False, EPI_S1MUL, beta_sp, thr_sp,
                             scale=1.0 / _SQRT2 if is_skip else 1.0, z=z_list[l - 1], g=zx[l - 1], nz=dh,
                             want_c=is_skip, out1=dst)
            if is_skip:
                de = ab[:, dh:] if de is None else de + ab[:, dh:]
        else:
            ab = gemm(zb, Ws[l], None, False, False)
            de = ab if de is None else de + ab
    gemm_group_tn(wgrad)
    colsum_into_multi(bgrad)
    d_e = de if need_e else None
    if shared is not None and not owner:
        # hand the sums to the owner's backward pass; autograd sees no weight gradient from this node
        shared["wgrad_acc"] = (dW, db, (tuple(bool(w) for w in need_w), tuple(bool(b) for b in need_b)))
        return (d_e, *([None] * (2 * L)))
    return (d_e, *dW, *db)


class _SdfMlpRows(torch.autograd.Function):
    """The node of _SdfMlp for a ROW RANGE of a batch another _SdfMlp node has just evaluated with the SAME weights:
    its forward results are that node's (out, g_e, z_l, a_l, v_l restricted to the rows), so nothing is recomputed; its
    backward is the same analytic pass on the row slices.  IDRNetwork.forward_static evaluates the ray points twice -
    once detached among the eikonal samples, once through SampleNetwork's re-parametrisation (the reference does it
    four times, implicit_differentiable_renderer.py:264,286,289,321-323) - and the two inputs are the same numbers:
    the second evaluation's 17 forward GEMMs disappear, its gradient flow (through its own embedding e2) is unchanged."""

    @staticmethod
    def forward(ctx, e2, cache, row0, n, *params):
        L = cache["meta"][0]
        sl = slice(int(row0), int(row0) + int(n))
        cp = lambda t: ops.dcopy_(torch.empty_like(t), t)    # noqa: E731  (kernel copy: no MEMCPY graph node)
        out = cp(cache["out"][sl])
        g_e = cp(cache["g_e"][sl])
        ctx.meta = cache["meta"]
        ctx.shared = cache            # weight / bias gradients are accumulated with the owner's (_sdf_mlp_backward)
        Ws = params[:L]
        if any(w.data_ptr() != cw.data_ptr() for w, cw in zip(Ws, cache["Ws"])):
            raise RuntimeError("hashmod: _SdfMlpRows must see the weights of the evaluation it reuses")
        ctx.save_for_backward(e2, cache["sdf"][sl], cache["c"][sl], cache["denom"][sl], *Ws,
                              *[a[sl] for a in cache["a_list"]], *[z[sl] for z in cache["z_list"]],
                              *[v[sl] for v in cache["v_list"]])
        return out, g_e

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out, d_ge):
        L = ctx.meta[0]
        grads = _sdf_mlp_backward(ctx.meta, ctx.saved_tensors, None, ctx.needs_input_grad[0],
                                  ctx.needs_input_grad[4:4 + L], ctx.needs_input_grad[4 + L:4 + 2 * L], d_out, d_ge,
                                  shared=ctx.shared, owner=False)
        return (grads[0], None, None, None, *grads[1:])


def sdf_mlp(e, weights, biases, skip_layer, beta_sp, thr_sp, beta_rho, cache_out=None):
    """(out [N, 1+fvs], d sdf/d e [N, E]); differentiable once w.r.t. e, weights and biases.
    cache_out: a dict that receives the evaluation's saved tensors (for sdf_mlp_rows)."""
    ops.require_gpu(e)
    return _SdfMlp.apply(e.contiguous(), int(skip_layer), float(beta_sp), float(thr_sp), float(beta_rho), cache_out,
                         *weights, *biases)


def sdf_mlp_rows(e2, cache, row0, n, weights, biases):
    """the same pair for rows [row0, row0 + n) of the batch `cache` was filled by (same weights, e2 == that batch's
    embedding rows in value): no forward work, own backward (through e2)."""
    ops.require_gpu(e2)
    if e2.shape[0] != n or row0 < 0 or row0 + n > cache["out"].shape[0]:
        raise ValueError("hashmod sdf_mlp_rows: row range does not match")
    return _SdfMlpRows.apply(e2.contiguous(), cache, int(row0), int(n), *weights, *biases)


class _ReluMlp(torch.autograd.Function):
    """Linear / ReLU stack of the rendering network (implicit_differentiable_renderer.py:211-221) as one
    first-order autograd node: forward GEMMs with the ReLU on their accumulators, backward dX GEMMs with the ReLU
    mask on theirs, weight gradients into buffers zeroed by one multi-tensor launch.  (The rendering network is
    differentiated once - by loss.backward(); the second-order path of IDR goes through its INPUT, the normals.)"""

    @staticmethod
    def forward(ctx, x, *params):
        L = len(params) // 2
        Ws, bs = params[:L], params[L:]
        acts = [x.contiguous()]
        for l in range(L - 1):
            acts.append(gemm_ep(acts[-1], Ws[l], bs[l], False, True, EPI_RELU, 0.0, 0.0, want_c=False)[1])
        y = gemm(acts[-1], Ws[L - 1], bs[L - 1], False, True)
        ctx.L = L
        ctx.save_for_backward(*Ws, *acts)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_y):
        L = ctx.L
        sv = ctx.saved_tensors
        Ws, acts = sv[:L], sv[L:]
        need_w = ctx.needs_input_grad[1:1 + L]
        need_b = ctx.needs_input_grad[1 + L:1 + 2 * L]
        dW = [torch.empty_like(Ws[l]) if need_w[l] else None for l in range(L)]
        db = [torch.empty(Ws[l].shape[0], dtype=torch.float32, device=d_y.device) if need_b[l] else None
              for l in range(L)]
        zero_list = [t for t in dW + db if t is not None]
        if zero_list:
            torch._foreach_zero_(zero_list)
        zb = d_y.contiguous()
        dx = None
        wgrad, bgrad = [], []
        for l in range(L - 1, -1, -1):
            if need_w[l]:
                wgrad.append((zb, acts[l], dW[l]))
            if need_b[l]:
                bgrad.append((zb, db[l]))
            if l > 0:     # h-bar_l = z-bar_l W_l, masked by relu'(z_{l-1}) = (h_l > 0)
                zb = gemm_ep(zb, Ws[l], None, False, False, EPI_RELUMASK, 0.0, 0.0, z=acts[l], want_c=False)[1]
            elif ctx.needs_input_grad[0]:
                dx = gemm(zb, Ws[0], None, False, False)
        gemm_group_tn(wgrad)
        colsum_into_multi(bgrad)
        return (dx, *dW, *db)


def relu_mlp(x, weights, biases):
    """Linear(+ReLU between layers) stack; differentiable once w.r.t. x, weights and biases."""
    ops.require_gpu(x)
    return _ReluMlp.apply(x, *weights, *biases)
