"""Ray-sharded data parallelism on the real model: two ranks (gloo backend, both on cuda:0 - RCCL refuses two ranks
per device) run the captured training step with the gradient all-reduce between the two graphs; a single process
fed the AVERAGED per-shard gradients must follow the same trajectory - parameters, loss and the ray tracer's
counters of rank 0 (the 4-rank rehearsal's `sdf_evals = 2 * rays` needed an explanation; the
non-finite-SDF counter of the tracer is asserted 0 on every rank and step here)."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_RAYS, STEPS, LR = 512, 5, 1.0e-4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build():
    import bench
    from helpers import idr_conf
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    torch.manual_seed(0)
    model = IDRNetwork(idr_conf("C1")).cuda()
    with torch.no_grad():  # let the hash features matter
        model.implicit_network.lin0.weight_v[:, 3:].normal_(0, 0.02)
        model.implicit_network.embed_model.embedder_obj.table.uniform_(-0.05, 0.05)
    model.train()
    return model, IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0), bench


def _shard_inputs(bench, rank):
    inp, gt = bench.synthetic_batch(1234 + rank, N_RAYS, "cuda")
    rs = np.random.RandomState(40 + rank)
    inp["object_mask"] = torch.from_numpy(rs.uniform(0, 1, (1, N_RAYS)) < 0.8).cuda()
    gt["rgb"] = torch.from_numpy(rs.uniform(-1, 1, (1, N_RAYS, 3)).astype(np.float32)).cuda()
    return inp, gt


def _worker(rank, world, port, out_dir, sparse=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HM_DIST_BACKEND="gloo")
    from hashmodnffbanks_idr_amd import parallel
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    parallel.init_distributed()
    torch.cuda.set_device(0)
    model, loss_fn, bench = _build()
    inp, gt = _shard_inputs(bench, rank)
    opt = ClipAdam(model.parameters(), lr=LR, max_norm=1.0)
    if sparse:   # the product's exchange: table gradients as (row, value) lists, device work inside the two graphs
        reducer = parallel.StaticGradExchange(model.parameters(),
                                              tables=[model.implicit_network.embed_model.embedder_obj,
                                                      model.rendering_network.embed_model.embedder_obj])
    else:        # dense all-reduce of every gradient
        reducer = parallel.GradAllReducer(model.parameters())
    stepper = GraphedTrainStep(model, loss_fn, opt, reducer, warmup=2)
    torch.manual_seed(100 + rank)
    rec = {"loss": [], "evals": [], "nonfinite": [], "unfinished": []}
    for _ in range(STEPS):
        _, lo = stepper.step(inp, gt)
        st = model.ray_tracer.last_stats
        rec["loss"].append(float(lo["loss"].item()))
        rec["evals"].append(int(st["sdf_evals"]))
        rec["nonfinite"].append(int(st["nonfinite"]))
        rec["unfinished"].append(int(st["unfinished"]))
    rec["graph"] = stepper.g_fb is not None
    if sparse:
        reducer.check()                                   # no touched row missed its payload
        torch.cuda.synchronize()
        for t in reducer.tables:                          # between steps the static dense gradients are all zero again
            assert int(torch.count_nonzero(t.dense)) == 0 and int(torch.count_nonzero(t.bits)) == 0
            assert int(t.count.item()) == 0
    rec["params"] = {n: p.detach().cpu() for n, p in model.named_parameters()}
    torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _single_process_reference():
    """the same two shards in ONE process: per step both shards' gradients, their mean, one clip + Adam step"""
    from hashmodnffbanks_idr_amd.model.loss import idr_loss_terms
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    model, loss_fn, bench = _build()
    shards = [_shard_inputs(bench, r) for r in range(2)]
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]    # the ranks' CPU generator streams
    opt = ClipAdam(model.parameters(), lr=LR, max_norm=1.0)
    params = [p for p in model.parameters() if p.requires_grad]
    rec = {"loss": [], "evals": []}
    bb = model.object_bounding_sphere
    for _ in range(STEPS):
        sums = None
        for r, (inp, gt) in enumerate(shards):
            steps = torch.empty(100).uniform_(0.0, 1.0, generator=gens[r]).cuda()
            eik = torch.empty(N_RAYS // 2, 3).uniform_(-bb, bb, generator=gens[r]).cuda()
            for p in params:
                p.grad = None
            out = model.forward_static(inp, eik, steps)
            lo = idr_loss_terms(out, gt["rgb"], loss_fn.eikonal_weight, loss_fn.mask_weight, loss_fn.alpha)
            lo["loss"].backward()
            if r == 0:
                rec["loss"].append(float(lo["loss"].item()))
                rec["evals"].append(int(model.ray_tracer.last_stats["sdf_evals"]))
            gs = [None if p.grad is None else p.grad.detach().clone() for p in params]
            sums = gs if sums is None else [a if b is None else (b if a is None else a + b) for a, b in zip(sums, gs)]
        for p, g in zip(params, sums):
            p.grad = None if g is None else g / 2.0
        opt.step()
    rec["params"] = {n: p.detach().cpu() for n, p in model.named_parameters()}
    return rec


@pytest.mark.parametrize("sparse", [False, True])
def test_two_gloo_ranks_follow_the_averaged_gradient_trajectory(sparse):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, sparse), nprocs=2, join=True)
        ranks = [torch.load(os.path.join(d, f"rank{r}.pt"), weights_only=False) for r in range(2)]
    ref = _single_process_reference()
    for r in ranks:
        assert r["graph"], "graph capture fell back to eager"
        assert r["nonfinite"] == [0] * STEPS and r["unfinished"] == [0] * STEPS
    # replicas stay identical across ranks (same averaged gradient, same update)
    rep = {n: (p - ranks[1]["params"][n]).abs().max().item() for n, p in ranks[0]["params"].items()}
    print("largest rank-0 / rank-1 parameter differences:", sorted(rep.items(), key=lambda kv: -kv[1])[:4])
    assert max(rep.values()) == 0.0, "replicas diverged"
    r0 = ranks[0]
    print("rank-0 sdf evals per step:", r0["evals"], " single process:", ref["evals"])
    print("rank-0 loss per step:", [f"{v:.6f}" for v in r0["loss"]], " single process:", [f"{v:.6f}" for v in ref["loss"]])
    for a, b in zip(r0["evals"], ref["evals"]):
        assert abs(a - b) <= 0.02 * b + 200, (r0["evals"], ref["evals"])
    for a, b in zip(r0["loss"], ref["loss"]):
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-6, (r0["loss"], ref["loss"])
    worst = 0.0
    for n, p in r0["params"].items():
        q = ref["params"][n]
        diff = (p - q).abs()
        worst = max(worst, diff.max().item())
        # Adam's first steps are lr * sign(g): an entry whose gradient is at noise level may go the other way
        assert diff.max().item() <= 2.5 * STEPS * LR, n
        assert diff.mean().item() <= 0.05 * LR, (n, diff.mean().item())
    print(f"max parameter difference after {STEPS} steps: {worst:.2e}")


def test_static_exchange_single_process_equals_plain_step():
    """with the table-gradient collector attached and NO process group the gradient must not be dropped -
    StaticGradExchange then applies the rank's own payload as the only list.  The captured step with the exchange must
    follow the captured step without it (same seeds; the two differ only in the order of fp32 atomic additions)."""
    from hashmodnffbanks_idr_amd import parallel
    from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    assert not torch.distributed.is_initialized()
    runs = []
    for with_exchange in (False, True):
        model, loss_fn, bench = _build()
        inp, gt = _shard_inputs(bench, 0)
        opt = ClipAdam(model.parameters(), lr=LR, max_norm=1.0)
        reducer = None
        if with_exchange:
            reducer = parallel.StaticGradExchange(model.parameters(),
                                                  tables=[model.implicit_network.embed_model.embedder_obj,
                                                          model.rendering_network.embed_model.embedder_obj])
        stepper = GraphedTrainStep(model, loss_fn, opt, reducer, warmup=2)
        t_init = model.implicit_network.embed_model.embedder_obj.table.detach().clone()
        torch.manual_seed(100)
        losses, norms = [], []
        for _ in range(STEPS):
            _, lo = stepper.step(inp, gt)
            losses.append(float(lo["loss"].item()))
            norms.append(float(opt.last_grad_norm.item()))
        assert stepper.g_fb is not None
        if with_exchange:
            reducer.check()
            emb = model.implicit_network.embed_model.embedder_obj
            assert emb.table.grad.data_ptr() == reducer.tables[0].dense.data_ptr()
            assert int(torch.count_nonzero(reducer.tables[0].dense)) == 0
        # the table was trained: Adam moves a row by ~lr per step once it has seen a gradient
        assert int(((model.implicit_network.embed_model.embedder_obj.table.detach() - t_init).abs() > 0.5 * LR).sum()) > 1000
        runs.append((losses, norms, {n: p.detach().clone() for n, p in model.named_parameters()}))
    (l0, n0, p0), (l1, n1, p1) = runs
    print("loss without / with exchange:", [f"{v:.6f}" for v in l0], [f"{v:.6f}" for v in l1])
    print("grad norm without / with exchange:", [f"{v:.6f}" for v in n0], [f"{v:.6f}" for v in n1])
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-4 * abs(a) + 1e-7
    for a, b in zip(n0, n1):
        assert abs(a - b) <= 1e-4 * abs(a) + 1e-7          # the table gradient is in the norm: a dropped one would show
    tab = "implicit_network.embed_model.embedder_obj.table"
    assert float((p1[tab] - p0[tab]).abs().max()) <= 2.5 * STEPS * LR
    assert float((p1[tab] - p0[tab]).abs().mean()) <= 0.05 * LR


def test_exchange_kernels_single_process():
    """device side of parallel.StaticGradExchange, kernel by kernel (csrc/hm_exchange.hip, hm_encode_bwd_table_tracked):
    the tracked scatter equals the plain scatter and lists every touched row exactly once; pack empties the dense
    gradient, the claim bits and (after clear) the counter; two ranks' lists applied in rank order give the mean of the
    two dense gradients bit for bit on repeated runs; the flat bucket copy is exact."""
    import ctypes as C
    import params as P
    from hashmodnffbanks_idr_amd import _lib, ops
    L, T, b, d = P.CONFIGS["C1"]
    res, rows = P.level_table(L, T, b, d)
    desc = ops.GridDesc(res, rows, 2)
    total = desc.total_rows
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(11)
    lists, denses = [], []
    for rank in range(2):
        n = 3000 + 17 * rank
        x = (torch.rand((n, 3), generator=g) * 2.2 - 1.1).to(dev)
        x[1000:1200] = x[:200].clone()                                  # duplicates: rows touched by several points
        df = torch.randn((n, L * 2), generator=g).to(dev)
        ref = ops.encode_bwd_table(desc, x, df, 0)
        dense = torch.zeros((total, 2), device=dev)
        bits = torch.zeros((total + 31) // 32, dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        cap = n * L
        rows_buf = torch.full((cap,), -7, dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        half = n // 2                                                   # two scatter calls share the tracking buffers
        for lo, hi in ((0, half), (half, n)):
            _lib.check(_lib.lib().hm_encode_bwd_table_tracked(
                desc.handle, _lib.dptr(x[lo:hi].contiguous()), hi - lo, _lib.dptr(df[lo:hi].contiguous()), L * 2,
                _lib.dptr(dense), 0, _lib.dptr(bits), _lib.dptr(count), _lib.dptr(rows_buf), cap, _lib.stream_ptr(x)))
        cnt = int(count.item())
        listed = rows_buf[:cnt].long()
        touched = torch.nonzero(ref.abs().sum(1) > 0).flatten()
        assert cnt <= cap and torch.unique(listed).numel() == cnt, "a row was listed twice"
        assert set(touched.tolist()) <= set(listed.tolist())            # (a touched row whose contributions cancel is listed too)
        assert (dense - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
        payload = torch.empty((cap, 3), dtype=torch.int32, device=dev)
        dense_before = dense.clone()
        _lib.check(_lib.lib().hm_rows_pack(_lib.dptr(dense), 2, _lib.dptr(rows_buf), _lib.dptr(count), cap, _lib.dptr(bits),
                                           _lib.dptr(payload), _lib.dptr(status), _lib.stream_ptr(x)))
        assert int(torch.count_nonzero(dense)) == 0 and int(torch.count_nonzero(bits)) == 0 and int(status.item()) == 0
        assert torch.equal(payload[:cnt, 0].long().sort().values, listed.sort().values)
        assert bool((payload[cnt:, 0] == -1).all())
        vals = payload[:cnt, 1:].view(torch.float32)
        assert torch.equal(vals, dense_before[payload[:cnt, 0].long()])
        lists.append(payload)
        denses.append(dense_before)
    cap = max(p.shape[0] for p in lists)
    gathered = torch.full((2, cap, 3), -1, dtype=torch.int32, device=dev)
    gathered[:, :, 1:] = 0
    for r, p in enumerate(lists):
        gathered[r, :p.shape[0]] = p
    outs = []
    for _ in range(2):
        acc = torch.zeros((total, 2), device=dev)
        _lib.check(_lib.lib().hm_rows_apply(_lib.dptr(acc), total, 2, _lib.dptr(gathered), cap, cap * 3, 2, 0.5,
                                            _lib.stream_ptr(acc)))
        outs.append(acc.clone())
    assert torch.equal(outs[0], outs[1])                                 # no atomics: bit-identical repeats
    assert torch.equal(outs[0], denses[0] * 0.5 + denses[1] * 0.5)       # rank order: 0 + a/2, then + b/2
    cnt_dev = torch.tensor([5], dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().hm_rows_clear(_lib.dptr(outs[0]), total, 2, _lib.dptr(gathered), cap, cap * 3, 2,
                                        _lib.dptr(cnt_dev), _lib.stream_ptr(acc)))
    assert int(torch.count_nonzero(outs[0])) == 0 and int(cnt_dev.item()) == 0
    # flat bucket
    srcs = [torch.randn(s, device=dev) for s in (1, 7, 4096, 8193, 512 * 512 + 3)]
    flat = torch.zeros(sum(t.numel() for t in srcs) + 1, device=dev)
    items = (_lib.CopyItem * len(srcs))()
    off = 1                                                               # (odd offset: the unaligned path)
    for i, t in enumerate(srcs):
        items[i] = _lib.CopyItem(t.data_ptr(), flat.data_ptr() + 4 * off, t.numel())
        off += t.numel()
    _lib.check(_lib.lib().hm_multi_copy_f32(C.cast(items, C.c_void_p), len(srcs), _lib.stream_ptr(flat)))
    assert torch.equal(flat[1:], torch.cat(srcs))
