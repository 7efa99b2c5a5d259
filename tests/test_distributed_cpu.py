"""world_size-2 gloo test of the ray-sharded data-parallel path (hashmodnffbanks_idr_amd/parallel.py)
on the CPU, with oracle/torch_ref.RefIDR standing in for the GPU model."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from helpers import idr_conf
    from oracle import torch_ref as R
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    torch.manual_seed(7)
    model = IDRNetwork(idr_conf("tiny", hidden=(64,) * 8, fvs=16, rdims=(64,) * 2)).cpu()
    with torch.no_grad():  # make the hash features matter
        model.implicit_network.lin0.weight_v[:, 3:].normal_(0, 0.05)
        model.implicit_network.embed_model.embedder_obj.table.uniform_(-0.3, 0.3)
    ref = R.RefIDR(model)
    ref.train()
    ref.ray_tracer.steps = torch.linspace(0.01, 0.99, 100)
    inp, gt = bench.synthetic_batch(5, 64, "cpu")
    rs = np.random.RandomState(3)
    inp["object_mask"] = torch.from_numpy(rs.uniform(0, 1, (1, 64)) < 0.8)
    inp["object_mask"][:, 48:] = False     # the last quarter of the rays has no surface: with 4 ranks, rank 3's shard
                                           # renders nothing and the rendering network gets no gradient there
    gt["rgb"] = torch.from_numpy(rs.uniform(-1, 1, (1, 64, 3)).astype(np.float32))
    return ref, IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0), inp, gt


def _shard_grads(ref, loss_fn, inp, gt, rank, world):
    from hashmodnffbanks_idr_amd import parallel
    mi, g = parallel.shard_rays(inp, gt, rank, world)
    torch.manual_seed(100 + rank)
    out = ref(mi)
    lo = loss_fn(out, g)
    ref.zero_grad()
    lo["loss"].backward()
    return lo


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from hashmodnffbanks_idr_amd import parallel
    r, w, _ = parallel.init_distributed(backend="gloo")
    ref, loss_fn, inp, gt = _build()
    lo = _shard_grads(ref, loss_fn, inp, gt, r, w)
    parallel.GradAllReducer(ref.parameters(), big_numel=256)()
    grads = {n: (None if p.grad is None else p.grad.clone().numpy()) for n, p in ref.named_parameters()}
    q.put((rank, float(lo["loss"]), grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])      # 4: rank 3 holds an empty-surface shard (no rendering-net gradient)
def test_mean_allreduce_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, loss, grads = q.get(timeout=500)
        res[r] = (loss, grads)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # single process: mean over the same two shards (same per-rank seeds)
    ref, loss_fn, inp, gt = _build()
    acc = {n: None for n, _ in ref.named_parameters()}
    losses = []
    for r in range(world):
        lo = _shard_grads(ref, loss_fn, inp, gt, r, world)
        losses.append(float(lo["loss"]))
        for n, p in ref.named_parameters():
            if p.grad is not None:
                acc[n] = p.grad.clone() if acc[n] is None else acc[n] + p.grad
    for r in range(world):
        assert abs(res[r][0] - losses[r]) <= 1e-6 * abs(losses[r])
        for n, g in res[r][1].items():
            if acc[n] is None:
                assert g is None
                continue
            want = (acc[n] / world).numpy()
            np.testing.assert_allclose(g, want, rtol=1e-6, atol=1e-6 * max(np.abs(want).max(), 1e-12), err_msg=n)
    # all ranks hold identical averaged gradients
    for r in range(1, world):
        for n in res[0][1]:
            if res[0][1][n] is not None:
                assert np.array_equal(res[0][1][n], res[r][1][n]), (r, n)
    if world == 4:      # the empty shard really was empty: rank 3 alone produced no rendering-network gradient
        ref3, loss_fn, inp, gt = _build()
        _shard_grads(ref3, loss_fn, inp, gt, 3, 4)
        assert all(p.grad is None for n, p in ref3.named_parameters() if n.startswith("rendering_network."))


def _sparse_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from hashmodnffbanks_idr_amd import parallel
    parallel.init_distributed(backend="gloo")
    rows, F = 5000, 2
    table = torch.nn.Parameter(torch.zeros(rows, F))
    other = torch.nn.Parameter(torch.zeros(7, 3))
    ex = parallel.RowValueExchange(table)
    red = parallel.GradAllReducer([table, other], big_numel=256, sparse=[ex])
    red.assume_dense = True
    out = []
    for step in range(3):                 # the same exchange object over several steps (static capacity)
        rs = np.random.RandomState(100 * step + rank)
        ex.begin_step()
        dense = torch.zeros(rows, F)
        if not (rank == world - 1 and step != 1):          # the last rank contributes nothing in steps 0 and 2
            for _ in range(2):                              # two contributions per step, rows repeat
                idx = torch.from_numpy(rs.randint(0, rows, 300))
                val = torch.from_numpy(rs.standard_normal((300, F)).astype(np.float32))
                ex.add(idx, val)
                dense.index_add_(0, idx, val)
        other.grad = torch.full((7, 3), float(rank + step))
        red()
        want = dense.clone()
        dist.all_reduce(want)                               # dense route of the same gradient
        want /= world
        out.append((table.grad.clone().numpy(), want.numpy(), other.grad.clone().numpy()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_sparse_row_value_exchange_equals_dense_mean(world):
    """parallel.RowValueExchange: all-gather of (row, value) pairs + local scatter == mean all-reduce of the dense
    gradient, including a rank with no contribution at all and repeated rows; the dense bucket next to it still works."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        for step, (got, want, other) in enumerate(res[r]):
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
            assert np.count_nonzero(want) > 0
            np.testing.assert_allclose(other, np.full((7, 3), np.mean([k + step for k in range(world)])), rtol=1e-6)
            assert np.array_equal(got, res[0][step][0])      # identical on every rank


def test_shard_rays_partition():
    from hashmodnffbanks_idr_amd import parallel
    inp = {"uv": torch.arange(24.).reshape(1, 12, 2), "object_mask": torch.ones(1, 12, dtype=torch.bool),
           "pose": torch.eye(4)[None], "intrinsics": torch.eye(4)[None]}
    gt = {"rgb": torch.arange(36.).reshape(1, 12, 3)}
    parts = [parallel.shard_rays(inp, gt, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p[0]["uv"] for p in parts], 1), inp["uv"])
    assert torch.equal(torch.cat([p[1]["rgb"] for p in parts], 1), gt["rgb"])
    with pytest.raises(ValueError):
        parallel.shard_rays(inp, gt, 0, 5)
