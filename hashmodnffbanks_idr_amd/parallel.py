"""Ray-sharded data parallelism: one process per GPU, torch.distributed over RCCL (xGMI).

The reference is single-GPU; rays are independent, so the pixel batch of one
iteration is partitioned across ranks and the only exchange is ONE gradient all-reduce (mean)
per step, issued after backward and before clip_grad_norm_ / Adam (idr_train.py:302-308 order).
With equal shards the mean of the per-rank gradients equals the gradient of the reference's
global loss (rgb / mask terms are sum/N, the eikonal term is a mean; loss.py:18,38,47).

Buckets are sized for xGMI (point-to-point, per-link bound): the hash-table gradient
(39.8 MiB at T=2^19, 223.5 MiB at T=2^22) is reduced in place as one large message; all MLP
gradients (~12 MiB) travel as one flat bucket.  Both collectives are issued asynchronously.
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
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # "nccl" IS RCCL on ROCm.  HM_DIST_BACKEND=gloo lets the N>1 path be rehearsed with several
            # ranks on ONE GPU (RCCL refuses two ranks per device).
            backend = os.environ.get("HM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
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


class GradAllReducer:
    """Mean all-reduce of every gradient: big tensors in place, the rest through one flat bucket."""

    def __init__(self, params, big_numel=1 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.big_numel = big_numel
        self.assume_dense = False   # set by GraphedTrainStep once the iteration is a replayed static graph
        self.frozen_grads = False   # set with it: the captured optimizer graph reads THESE .grad tensors; never rebind
        self._flat = None
        self._flat_key = None
        self._views = None

    def __call__(self):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        dev = self.params[0].device
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
