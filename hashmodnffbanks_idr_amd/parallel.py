"""Ray-sharded data parallelism: one process per GPU, torch.distributed over RCCL (xGMI).

The reference is single-GPU; rays are independent, so the pixel batch of one
iteration is partitioned across ranks and the only exchange is ONE gradient all-reduce (mean)
per step, issued after backward and before clip_grad_norm_ / Adam (idr_train.py:302-308 order).
With equal shards the mean of the per-rank gradients equals the gradient of the reference's
global loss (rgb / mask terms are sum/N, the eikonal term is a mean; loss.py:18,38,47).

Buckets are sized for xGMI (point-to-point, per-link bound): all MLP gradients (~12 MiB) travel as one flat
bucket.  The hash-table gradient (39.8 MiB at T=2^19, 223.5 MiB at T=2^22) has two routes:
  * dense: reduced in place as one large message (always available);
  * sparse (static / graph-captured steps, StaticGradExchange + TouchedRowExchange): a step touches <= 0.4 % of the
    rows.  The table backward lists the rows it touches, the rank's (row, value) pairs (~1 MB) are all-gathered and
    every rank adds all ranks' lists in rank order - no sort, no vendor library, every device-side piece inside the
    step's two captured graphs, exactly two collectives per step.  RowValueExchange is the eager, CPU-capable form of
    the same idea (gloo tests).  The result equals the dense mean up to the order of the fp32 additions
    (tests/test_distributed_cpu.py, tests/test_distributed_gpu.py).
Why not "overlap the table all-reduce with backward": the embedding is the FIRST layer, so its gradient is the LAST
thing backward produces - there is nothing left to overlap with, and clip_grad_norm_ needs every gradient before
the first Adam update.  Shrinking the message (sparse route) is what removes the cost.
Works unchanged with the gloo backend on CPU tensors (tests).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HM_DIST_FORCE=1: create the process group for a single rank too (rehearses the RCCL code path on one GPU)
    if (world > 1 or os.environ.get("HM_DIST_FORCE") == "1") and not dist.is_initialized():
        if backend is None:
            # "nccl" IS RCCL on ROCm.  HM_DIST_BACKEND=gloo lets the N>1 path be rehearsed with several
            # ranks on ONE GPU (RCCL refuses two ranks per device).
            backend = os.environ.get("HM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_rays(model_input, ground_truth, rank, world):
    """Equal contiguous shards of the pixel dimension (uv [B,N,2], object_mask [B,N], rgb [B,N,3])."""
    n = model_input["uv"].shape[1]
    if n % world != 0:
        raise ValueError(f"number of rays ({n}) must be divisible by the world size ({world})")
    per = n // world
    sl = slice(rank * per, (rank + 1) * per)
    mi = dict(model_input)
    mi["uv"] = model_input["uv"][:, sl]
    mi["object_mask"] = model_input["object_mask"][:, sl]
    gt = dict(ground_truth)
    gt["rgb"] = ground_truth["rgb"][:, sl]
    return mi, gt


class _SparseExchange:
    """Common part of the sparse gradient exchanges: a fixed-capacity payload [cap, width] per rank (rows beyond the
    step's contributions stay zero and add nothing), all-gathered, then applied by every rank to a STATIC dense
    gradient tensor that is (re)attached to the parameter - so a captured optimizer graph keeps reading one address."""

    def __init__(self, param, width):
        self.param, self.width = param, width
        self.active = True
        self.payload = None          # [cap, width], allocated by the first step
        self.cursor = 0
        self.dense = None            # static dense gradient of `param`
        self._gathered = None

    def begin_step(self):
        self.cursor = 0

    def _rows(self, n):
        """rows [cursor, cursor+n) of the payload (grown only while no capacity has been fixed by an exchange)"""
        dev = self.param.device
        if self.payload is None or self.cursor + n > self.payload.shape[0]:
            if self._gathered is not None:
                raise RuntimeError("sparse gradient exchange: more contributions than the capacity fixed by the first "
                                   "exchange (the step is not static)")
            new = torch.zeros((max(2 * (self.cursor + n), 1024), self.width), dtype=torch.float32, device=dev)
            if self.payload is not None and self.cursor:
                new[:self.cursor].copy_(self.payload[:self.cursor])
            self.payload = new
        view = self.payload[self.cursor:self.cursor + n]
        self.cursor += n
        return view

    def exchange(self, world):
        dev = self.param.device
        if self._gathered is None:
            # first exchange: the ranks agree on ONE capacity = the largest row count of this step (static steps repeat
            # it exactly; a rank with fewer rows pads with zero rows, which add nothing).  No slack is kept: padding
            # rows all land on one table row per level and would serialise the deterministic scatter's run sum.
            need = torch.tensor([max(self.cursor, 1)], device=dev, dtype=torch.int64)
            dist.all_reduce(need, op=dist.ReduceOp.MAX)
            cap = int(need)
            fixed = torch.zeros((cap, self.width), dtype=torch.float32, device=dev)
            if self.payload is not None and self.cursor:
                fixed[:self.cursor].copy_(self.payload[:self.cursor])
            self.payload = fixed
            self._gathered = torch.empty((world * cap, self.width), dtype=torch.float32, device=dev)
        elif self.cursor < self.payload.shape[0]:
            self.payload[self.cursor:].zero_()          # stale rows of an earlier, larger step must add nothing
        cap = self.payload.shape[0]
        if dist.get_backend() == "nccl":      # RCCL: one all-gather straight into the flat buffer
            dist.all_gather_into_tensor(self._gathered, self.payload)
        else:
            dist.all_gather(list(self._gathered.split(cap, 0)), self.payload)
        if self.dense is None or self.dense.shape != self.param.shape or self.dense.device != self.param.device:
            self.dense = torch.zeros_like(self.param)
        else:
            self.dense.zero_()
        self.apply(self._gathered, self.dense)
        self.dense.mul_(1.0 / world)
        self.param.grad = self.dense

    def attach(self):
        """parameter.grad = the static dense tensor (after zero_grad(set_to_none=True), before a graph capture)"""
        if self.dense is None:
            self.dense = torch.zeros_like(self.param)
        self.param.grad = self.dense


class RowValueExchange(_SparseExchange):
    """(row id, value) pairs of a [rows, F] parameter: payload row = [row id as float64-safe int in fp32 pairs, F values].
    Row ids are split into two fp32-exact halves (hi = id // 65536, lo = id % 65536) so the payload stays one fp32
    tensor for a single collective."""

    def __init__(self, param):
        super().__init__(param, 2 + param.shape[1])

    def add(self, rows, values):
        view = self._rows(rows.shape[0])
        view[:, 0] = (rows // 65536).to(torch.float32)
        view[:, 1] = (rows % 65536).to(torch.float32)
        view[:, 2:] = values

    def apply(self, gathered, dense):
        rows = gathered[:, 0].to(torch.int64) * 65536 + gathered[:, 1].to(torch.int64)
        dense.index_add_(0, rows, gathered[:, 2:])      # zero-valued padding rows add nothing (row 0)


class TouchedRowExchange:
    """Hash-table gradient of one MultiResHashGridMLP in the static (graph-captured) data-parallel step.

    Registered as ``emb.grad_collector``: the encoder's autograd node (ops._HashFeatures.backward) hands its
    (points, feature gradients) to add(), which scatters them straight into the STATIC dense gradient tensor (bound as
    ``table.grad``, all zero between steps) with hm_encode_bwd_table_tracked - the ordinary atomic scatter that also
    lists every row it touches once.  StaticGradExchange then packs the listed rows into (row, value) pairs, all-gathers
    them and lets every rank add ALL ranks' lists in rank order (hm_rows_apply): a step touches <= 0.4 % of the rows, the
    message is ~1 MB per rank instead of 40 / 224 MB, nothing is sorted, and the replicas stay bitwise identical because
    every row's sum runs in the same order everywhere (a rank's own partial sums are computed once, by that rank)."""

    def __init__(self, emb):
        if emb.frac_mode != "reference":
            # the trilinear encoder sits on the x-graph: autograd.grad(e, x, create_graph=True) runs its table backward
            # too (a custom Function cannot see that only d/dx was asked for) and would scatter bogus contributions;
            # its second-order table term (hm_encode_bwd_table_jvp) is dense as well
            raise NotImplementedError("TouchedRowExchange: frac_mode='trilinear' uses the dense table all-reduce")
        self.emb, self.param = emb, emb.table
        self.F = int(emb.n_features)
        self.total_rows = int(emb.desc.total_rows)
        self.active = True
        self.frozen = False
        self.cap = 0                 # listed rows per step (upper bound: points x levels of all scatter calls)
        self.cursor = 0              # host-side count of this step's contributions
        dev = emb.table.device
        self.dense = torch.zeros_like(emb.table)
        self.bits = torch.zeros((self.total_rows + 31) // 32, dtype=torch.int32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.rows = None
        emb.grad_collector = self

    def begin_step(self):
        self.cursor = 0

    def attach(self):
        """table.grad = the static dense tensor (after zero_grad(set_to_none=True), before a graph capture)"""
        self.param.grad = self.dense

    def _ensure(self, need):
        if self.rows is not None and need <= self.cap:
            return
        if self.frozen:
            raise RuntimeError("TouchedRowExchange: more table-gradient contributions than the captured step has room "
                               "for (the step is not static)")
        new = torch.empty(max(need, 1024), dtype=torch.int32, device=self.param.device)
        if self.rows is not None and self.cap:
            new[:self.cap].copy_(self.rows[:self.cap])
        self.rows, self.cap = new, int(new.numel())

    def add(self, x, d_feat):
        from . import _lib
        x = x.detach().reshape(-1, 3)
        d_feat = d_feat.detach()
        if d_feat.stride(-1) != 1:
            d_feat = d_feat.contiguous()
        n = x.shape[0]
        if self.param.grad is None:
            self.param.grad = self.dense
        elif self.param.grad.data_ptr() != self.dense.data_ptr():
            raise RuntimeError("TouchedRowExchange: table.grad was re-bound; call attach() after zero_grad()")
        self._ensure(self.cursor + n * self.emb.n_levels)
        _lib.check(_lib.lib().hm_encode_bwd_table_tracked(
            self.emb.desc.handle, _lib.dptr(x), n, _lib.dptr(d_feat), d_feat.stride(0), _lib.dptr(self.dense), 0,
            _lib.dptr(self.bits), _lib.dptr(self.count), _lib.dptr(self.rows), self.cap, _lib.stream_ptr(x)))
        self.cursor += n * self.emb.n_levels


class StaticGradExchange:
    """Gradient exchange of the static training step (training.graph_step.GraphedTrainStep): device work as kernels that
    are captured into the step's two graphs, and exactly TWO collectives between them.

        g_fb  : forward, loss, backward (table scatters through TouchedRowExchange), then pack():
                  hm_rows_pack     per hash table: listed rows -> this rank's (row, value) payload
                  hm_multi_copy    every other gradient -> one flat fp32 bucket (one launch)
        eager : communicate():  all_gather(payload)  +  all_reduce(bucket, AVG)         [RCCL over xGMI]
        g_opt : apply(): hm_rows_apply (one launch per rank: dense table gradient = mean of all ranks' lists),
                clip_grad_norm_ + Adam reading the averaged gradients straight from the bucket (no copy back),
                finish(): hm_rows_clear (dense table gradient back to zero)

    Order as in the reference runner (idr_train.py:302-308): backward -> [exchange] -> clip_grad_norm_ -> Adam.
    Without an initialised process group (single process) communicate() moves nothing and the payload is applied as
    the only list - the table gradient is never dropped."""

    def __init__(self, params, tables=(), optimizer=None):
        self.tables = [t if isinstance(t, TouchedRowExchange) else TouchedRowExchange(t) for t in tables]
        skip = {id(t.param) for t in self.tables}
        self.params = [p for p in params if p.requires_grad and id(p) not in skip]
        self.opt = optimizer
        self.frozen = False
        self._layout = None
        self.flat = self.views = self.payload = self.gathered = None
        self._live = None
        self._checked = False

    # -- protocol ---------------------------------------------------------------------------------------
    def begin_step(self):
        for t in self.tables:
            t.begin_step()

    def attach(self):
        for t in self.tables:
            t.attach()

    def _world(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    def _build_layout(self, key, live):
        if self.frozen:
            raise RuntimeError("StaticGradExchange: the set of gradients changed after the step was captured")
        dev = (live[0] if live else self.tables[0].param).device
        world = self._world()
        self.flat = torch.zeros(max(sum(p.numel() for p in live), 1), dtype=torch.float32, device=dev)
        self.views = [v.view_as(p) for v, p in zip(self.flat.split([p.numel() for p in live]), live)] if live else []
        offs, total = [], 0
        for t in self.tables:
            t._ensure(max(t.cursor, 1))
            t.cap = max(t.cursor, 1)          # exactly this step's contribution count: static steps repeat it
            offs.append(total)
            total += t.cap * (1 + t.F)
        self.offs, self.stride = offs, max(total, 1)
        self.payload = torch.full((self.stride,), -1, dtype=torch.int32, device=dev)
        self.gathered = (torch.empty((world, self.stride), dtype=torch.int32, device=dev) if dist.is_initialized()
                         else self.payload.view(1, -1))
        self._layout, self._live = key, live
        self._checked = False
        if self.opt is not None:
            self.opt.grad_override = {p: v for p, v in zip(live, self.views)}

    def pack(self):
        """after backward: payloads and the flat bucket (kernels only; captured at the end of g_fb)"""
        from . import _lib
        live = [p for p in self.params if p.grad is not None]
        key = (tuple(id(p) for p in live), tuple(t.cursor for t in self.tables))
        if key != self._layout:
            self._build_layout(key, live)
        for t, off in zip(self.tables, self.offs):
            _lib.check(_lib.lib().hm_rows_pack(_lib.dptr(t.dense), t.F, _lib.dptr(t.rows), _lib.dptr(t.count), t.cap,
                                               _lib.dptr(t.bits), self.payload.data_ptr() + 4 * off, _lib.dptr(t.status),
                                               _lib.stream_ptr(t.dense)))
        if live:
            items = (_lib.CopyItem * len(live))()
            for i, (p, v) in enumerate(zip(live, self.views)):
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise RuntimeError("StaticGradExchange: contiguous fp32 gradients expected")
                items[i] = _lib.CopyItem(g.data_ptr(), v.data_ptr(), g.numel())
            import ctypes as C
            _lib.check(_lib.lib().hm_multi_copy_f32(C.cast(items, C.c_void_p), len(live), _lib.stream_ptr(self.flat)))

    def communicate(self):
        """the two collectives (eager, between the two graphs)"""
        if not dist.is_initialized():
            return
        world = dist.get_world_size()
        if not self._checked:
            # once per layout: every rank must run the same static step (same bucket and payload sizes)
            sig = torch.tensor([self.flat.numel(), self.stride, -self.flat.numel(), -self.stride], device=self.flat.device,
                               dtype=torch.int64)
            dist.all_reduce(sig, op=dist.ReduceOp.MAX)
            sig = sig.tolist()
            if sig[0] != -sig[2] or sig[1] != -sig[3]:
                raise RuntimeError("StaticGradExchange: the ranks' static steps differ in size "
                                   f"(bucket {sig[0]} vs {-sig[2]}, payload {sig[1]} vs {-sig[3]})")
            self._checked = True
        nccl = dist.get_backend() == "nccl"
        if nccl:      # RCCL: one all-gather straight into the [world, stride] buffer, one averaging all-reduce
            dist.all_gather_into_tensor(self.gathered.view(-1), self.payload)
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_gather(list(self.gathered.unbind(0)), self.payload)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / world)

    def apply(self):
        """before clip + Adam: dense table gradients from all ranks' lists (captured at the start of g_opt)"""
        from . import _lib
        world = self.gathered.shape[0]
        for t, off in zip(self.tables, self.offs):
            _lib.check(_lib.lib().hm_rows_apply(_lib.dptr(t.dense), t.total_rows, t.F, self.gathered.data_ptr() + 4 * off,
                                                t.cap, self.stride, world, 1.0 / world, _lib.stream_ptr(t.dense)))

    def finish(self):
        """after the optimizer step: touched rows of the dense table gradients back to zero"""
        from . import _lib
        world = self.gathered.shape[0]
        for t, off in zip(self.tables, self.offs):
            _lib.check(_lib.lib().hm_rows_clear(_lib.dptr(t.dense), t.total_rows, t.F, self.gathered.data_ptr() + 4 * off,
                                                t.cap, self.stride, world, _lib.dptr(t.count), _lib.stream_ptr(t.dense)))

    def freeze(self):
        self.frozen = True
        for t in self.tables:
            t.frozen = True

    def check(self):
        """host read (synchronises): raises when a step claimed more rows than the payload holds"""
        for t in self.tables:
            over = int(t.status.item())
            if over:
                raise RuntimeError(f"TouchedRowExchange: {over} touched rows did not fit the payload (cap {t.cap})")


class GradAllReducer:
    """Mean all-reduce of every gradient: big tensors in place, the rest through one flat bucket; parameters
    listed in `sparse` (exchange objects) take the sparse route instead."""

    def __init__(self, params, big_numel=1 << 20, sparse=()):
        self.sparse = list(sparse)
        skip = {id(ex.param) for ex in self.sparse}
        self.params = [p for p in params if p.requires_grad and id(p) not in skip]
        self.big_numel = big_numel
        self.assume_dense = False   # set by GraphedTrainStep once the iteration is a replayed static graph
        self.frozen_grads = False   # set with it: the captured optimizer graph reads THESE .grad tensors; never rebind
        self._flat = None
        self._flat_key = None
        self._views = None

    def __call__(self):
        if not dist.is_initialized() or (dist.get_world_size() == 1 and os.environ.get("HM_DIST_FORCE") != "1"):
            return
        world = dist.get_world_size()
        dev = self.params[0].device
        for ex in self.sparse:
            ex.exchange(world)
        # a parameter may have no gradient on one rank only (e.g. no surface hit in its shard): agree first
        if self.assume_dense and all(p.grad is not None for p in self.params):
            have = [1] * len(self.params)      # replayed static iteration: every rank has every gradient, no handshake
        else:
            have = torch.tensor([0 if p.grad is None else 1 for p in self.params], device=dev, dtype=torch.int32)
            dist.all_reduce(have, op=dist.ReduceOp.MAX)
            have = have.tolist()
        big, small = [], []
        for p, h in zip(self.params, have):
            if not h:
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            (big if p.numel() >= self.big_numel else small).append(p)
        works = []
        for p in big:
            if not p.grad.is_contiguous():
                if self.frozen_grads:
                    raise RuntimeError("GradAllReducer: a gradient became non-contiguous after the step was captured; "
                                       "rebinding .grad would detach the captured optimizer graph from it")
                p.grad = p.grad.contiguous()
            works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True))
        if small:
            key = tuple((p.data_ptr(), p.numel()) for p in small)
            if self._flat is None or self._flat_key != key or self._flat.device != dev:
                self._flat = torch.empty(sum(p.numel() for p in small), dtype=torch.float32, device=dev)
                self._views = [v.view_as(p) for v, p in zip(self._flat.split([p.numel() for p in small]), small)]
                self._flat_key = key
            grads = [p.grad for p in small]
            torch._foreach_copy_(self._views, grads)          # one multi-tensor launch instead of one copy per tensor
            works.append(dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        inv = 1.0 / world
        if big:
            torch._foreach_mul_([p.grad for p in big], inv)
        if small:
            self._flat.mul_(inv)
            torch._foreach_copy_(grads, self._views)


def train_step(model, loss_fn, optimizer, model_input, ground_truth, reducer=None, max_norm=1.0):
    """One iteration in the reference runner's order (idr_train.py:294-308): forward, loss, backward,
    [gradient all-reduce], clip_grad_norm_(1.0), dense Adam step."""
    out = model(model_input)
    loss_out = loss_fn(out, ground_truth)
    optimizer.zero_grad()
    loss_out["loss"].backward()
    if reducer is not None:
        reducer()
    if not getattr(optimizer, "fused_clip", False):   # training.optim.ClipAdam clips inside its own pass
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    return out, loss_out
