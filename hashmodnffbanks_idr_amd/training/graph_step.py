"""One training iteration as a captured HIP graph (the MI355X answer to the reference runner's
~2000 eager launches and >= 12 host synchronisations per iteration, training/idr_train.py:278-321).

The iteration keeps the reference's order - forward, IDRLoss, backward, [gradient all-reduce],
clip_grad_norm_(1.0), dense Adam - but runs the static-shape forward (IDRNetwork.forward_static:
masks instead of boolean gathers, the whole ray search enqueued by one C call) so that nothing
depends on a device->host read.  The first calls run eagerly (warm-up on a side stream), then the
kernels of [forward + loss + backward] and of [clip + Adam] are captured once with
torch.cuda.CUDAGraph (hipGraph underneath) and replayed; only the two CPU-generator draws the reference
makes per iteration (eikonal samples, closest-approach fractions) are copied in before each replay.
With more than one rank the gradient exchange sits between the two graphs: parallel.StaticGradExchange has its device
work (payload packing, flat bucket, dense-gradient assembly) captured INTO the two graphs, so only its two collectives
run eagerly; a plain parallel.GradAllReducer (dense all-reduce) runs eagerly as a whole.
"""
import os

import torch

from .. import _lib
from ..model.loss import idr_loss_terms


class LocalTableGrad:
    """``emb.grad_collector`` of the single-process static step: every table backward of an iteration (the eikonal /
    ray-point evaluation, the re-parametrised ray points, a view-direction grid) scatters into ONE dense gradient that is
    bound as ``table.grad`` and zeroed once per step.  Plain autograd gives each scatter node a ``zeros_like(table)`` of
    its own and adds the dense results up: 3 fills + 2 adds over 40 MB (T = 2^19) / 224 MB (T = 2^22) per step.
    Same protocol as parallel.TouchedRowExchange (begin_step / attach / add), without the row lists; reference frac
    mode only, for the reason given there."""

    def __init__(self, emb):
        if emb.frac_mode != "reference":
            raise NotImplementedError("LocalTableGrad: frac_mode='trilinear' keeps autograd's dense accumulation")
        self.emb, self.param = emb, emb.table
        self.active = False
        self.dense = torch.zeros_like(emb.table)
        emb.grad_collector = self

    def begin_step(self):
        self.active = True          # (only while the stepper's own forward / backward runs: other users of the model -
        self.dense.zero_()          #  eager steps, tests - keep autograd's table gradient)   [zero_ = a fill kernel]

    def end_step(self):
        self.active = False

    def attach(self):
        self.param.grad = self.dense

    def add(self, x, d_feat):
        from .. import ops
        if self.param.grad is None:
            self.param.grad = self.dense
        elif self.param.grad.data_ptr() != self.dense.data_ptr():
            raise RuntimeError("LocalTableGrad: table.grad was re-bound; call attach() after zero_grad()")
        ops.encode_bwd_table(self.emb.desc, x.detach().reshape(-1, 3), d_feat.detach(), 0, out=self.dense)

    def release(self):
        if self.emb.grad_collector is self:
            self.emb.grad_collector = None


def local_table_grads(model):
    """LocalTableGrad for every reference-mode hash grid of the model that has no gradient collector yet"""
    out = []
    for net in (model.implicit_network, model.rendering_network):
        emb = getattr(getattr(net, "embed_model", None), "embedder_obj", None)
        emb = getattr(emb, "grid_enc", emb)          # filter-bank embedders own a hash grid too
        if (emb is not None and hasattr(emb, "grad_collector") and emb.grad_collector is None
                and getattr(emb, "frac_mode", None) == "reference" and emb.table.requires_grad):
            out.append(LocalTableGrad(emb))
    return out


class GraphedTrainStep:
    def __init__(self, model, loss_fn, optimizer, reducer=None, max_norm=1.0, warmup=3, use_graph=True,
                 sync_each_step=None):
        self.model, self.loss_fn, self.opt, self.reducer = model, loss_fn, optimizer, reducer
        # single process: the hash tables' gradients accumulate in one static dense tensor each (LocalTableGrad);
        # HM_LOCAL_TABLE_GRAD=0 keeps autograd's per-node dense gradients (A/B)
        self.local_tables = (local_table_grads(model)
                             if reducer is None and os.environ.get("HM_LOCAL_TABLE_GRAD", "1") != "0" else [])
        self.max_norm, self.warmup_left, self.use_graph = max_norm, warmup, use_graph
        self.static_exchange = reducer is not None and hasattr(reducer, "pack")   # parallel.StaticGradExchange
        if self.static_exchange and reducer.opt is None:
            reducer.opt = optimizer      # it points the optimizer at the flat, all-reduced gradient bucket
        if sync_each_step is None:
            # default: replays run ahead of the host.  The round-1 replay fault came from MEMSET / MEMCPY graph nodes;
            # the captured iteration now holds kernel nodes only (tests/test_graph_step_gpu.py asserts it), and
            # 2 x 600 no-sync steps at the bench configuration ran clean.  With a reducer whose device work
            # is NOT captured (GradAllReducer: copies, memsets and allocator traffic run eagerly between the replays -
            # the pattern of the round-1 fault) the per-step synchronisation stays on; StaticGradExchange leaves only
            # its two collectives between the graphs (600-step run-ahead rehearsal: DESIGN.md section 6).
            # HM_GRAPH_SYNC=1 / 0 overrides either way.
            env = os.environ.get("HM_GRAPH_SYNC")
            sync_each_step = (env == "1") if env is not None else (reducer is not None and not self.static_exchange)
        self.sync_each_step = sync_each_step
        self.g_fb = self.g_opt = None
        self.side = None
        self._stage = None
        self.static = None
        self.out = self.loss_out = None
        for grp in optimizer.param_groups:
            if use_graph and not (grp.get("capturable", False) or getattr(optimizer, "fused_clip", False)):
                raise ValueError("GraphedTrainStep needs training.optim.ClipAdam or torch.optim.Adam(..., capturable=True)")

    # -- pieces -------------------------------------------------------------------------------
    def _draws(self, n_rays, dev):
        """the reference's two CPU-generator draws of an iteration, in its order (ray_tracing.py:277 first,
        then implicit_differentiable_renderer.py:279).

        They are written into PINNED staging buffers owned by this object (a ring of two, each guarded by an
        event): the host->device copies are asynchronous, and a copy whose pageable source tensor has
        already been freed by the time the copy engine runs is a GPU memory-access fault."""
        rt = self.model.ray_tracer
        if self._stage is None:
            self._stage = [(torch.empty(rt.n_steps).pin_memory(), torch.empty(n_rays // 2, 3).pin_memory(),
                            torch.cuda.Event()) for _ in range(2)]
            self._slot = 0
        steps, eik, ev = self._stage[self._slot]
        self._slot ^= 1
        ev.synchronize()          # the copy that last read this slot has completed
        steps.uniform_(0.0, 1.0)
        bb = self.model.object_bounding_sphere
        eik.uniform_(-bb, bb)
        return steps, eik, ev

    def _sparse(self):
        return getattr(self.reducer, "sparse", ()) if self.reducer is not None else self.local_tables

    def _fwd_bwd(self):
        for ex in self._sparse():
            ex.begin_step()
        if self.static_exchange:
            self.reducer.begin_step()
        s = self.static
        out = self.model.forward_static(s["input"], s["eik"], s["steps"])
        lo = idr_loss_terms(out, s["rgb"], self.loss_fn.eikonal_weight, self.loss_fn.mask_weight, self.loss_fn.alpha)
        lo["loss"].backward()
        for ex in self.local_tables:
            ex.end_step()
        if self.static_exchange:
            self.reducer.pack()          # payloads + flat bucket: kernels, part of the captured forward/backward graph
        return out, lo

    def _exchange(self):
        """what runs between the two graphs"""
        if self.reducer is None:
            return
        if self.static_exchange:
            self.reducer.communicate()   # the two collectives, nothing else
        else:
            self.reducer()

    def _update(self):
        if self.static_exchange:
            self.reducer.apply()         # dense table gradients from all ranks' lists (captured)
        if not getattr(self.opt, "fused_clip", False):   # training.optim.ClipAdam clips inside its own pass
            if self.static_exchange:
                raise ValueError("GraphedTrainStep: StaticGradExchange needs training.optim.ClipAdam (it reads the "
                                 "averaged gradients from the flat bucket)")
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.max_norm)
        self.opt.step()
        if self.static_exchange:
            self.reducer.finish()

    def _eager_iteration(self):
        self.opt.zero_grad(set_to_none=True)
        for ex in self.local_tables:
            ex.attach()
        if self.static_exchange:
            self.reducer.attach()
        out, lo = self._fwd_bwd()
        # keep VALUES only: a live autograd graph would keep this iteration's AccumulateGrad nodes (bound to
        # this stream) alive, and a later capture on another stream would then leave the gradient
        # accumulation outside the captured graph
        self.out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        self.loss_out = {k: v.detach() for k, v in lo.items()}
        del out, lo
        self._exchange()
        self._update()

    # -- driver -------------------------------------------------------------------------------
    def step(self, model_input, ground_truth):
        dev = model_input["uv"].device
        n_rays = model_input["uv"].shape[0] * model_input["uv"].shape[1]
        steps, eik, ev = self._draws(n_rays, dev)
        if self.static is None:
            self.static = {"input": {k: v.to(dev).clone() for k, v in model_input.items()},
                           "rgb": ground_truth["rgb"].to(dev).clone(),
                           "eik": torch.empty(eik.shape, device=dev), "steps": torch.empty(steps.shape, device=dev)}
        s = self.static
        for k, v in model_input.items():
            if v.data_ptr() != s["input"][k].data_ptr():
                s["input"][k].copy_(v, non_blocking=v.is_cuda)   # host sources are copied synchronously
        s["rgb"].copy_(ground_truth["rgb"], non_blocking=ground_truth["rgb"].is_cuda)
        s["eik"].copy_(eik, non_blocking=True)
        s["steps"].copy_(steps, non_blocking=True)
        ev.record()

        if not self.use_graph or self.warmup_left > 0:
            self.warmup_left -= 1
            if self.use_graph:
                # warm-up iterations must run on a side stream before capture (autograd's backward
                # thread and the library workspaces otherwise stay bound to the legacy stream and
                # hipStreamEndCapture crashes)
                if self.side is None:
                    self.side = torch.cuda.Stream()
                self.side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.side):
                    self._eager_iteration()
                torch.cuda.current_stream().wait_stream(self.side)
            else:
                self._eager_iteration()
            return self.out, self.loss_out

        if self.g_fb is None:
            # capture (records the kernels, does not run them), then fall through to the first replay
            self.model.implicit_network._force_repack = True  # make the capture contain the weight re-pack
            self.opt.zero_grad(set_to_none=True)
            for ex in self._sparse():      # their dense gradients are static tensors the optimizer graph reads
                ex.attach()
            if self.static_exchange:
                self.reducer.attach()
            torch.cuda.synchronize()
            # with RCCL alive its watchdog thread polls events; only this thread's calls must obey capture rules
            mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
            if mode == "thread_local":
                # let the watchdog retire the warm-up iterations' collectives (it sweeps every 100 ms) before this stream
                # starts capturing: a sweep that still holds one of their events during the capture died with
                # hipErrorCapturedEvent once under rocprofv3 (gpurun_out/r3ac/prof_rccl1.log) - a capture failure on a
                # data-parallel rank is fatal
                import time
                time.sleep(0.5)
            self.out = self.loss_out = None
            if self.side is None:
                self.side = torch.cuda.Stream()
            try:
                dump = os.environ.get("HM_GRAPH_DUMP")       # directory for hipGraphDebugDotPrint output (debugging)
                g_fb = torch.cuda.CUDAGraph(keep_graph=True) if dump else torch.cuda.CUDAGraph()
                if dump:
                    g_fb.enable_debug_mode()
                with torch.cuda.graph(g_fb, stream=self.side, capture_error_mode=mode):   # same stream as the warm-up
                    out, lo = self._fwd_bwd()
                    self.out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
                    self.loss_out = {k: v.detach() for k, v in lo.items()}
                    del out, lo
                g_opt = torch.cuda.CUDAGraph(keep_graph=True) if dump else torch.cuda.CUDAGraph()
                if dump:
                    g_opt.enable_debug_mode()
                with torch.cuda.graph(g_opt, stream=self.side, capture_error_mode=mode):
                    self._update()
                self.g_fb, self.g_opt = g_fb, g_opt
                if self.reducer is not None and hasattr(self.reducer, "assume_dense"):
                    self.reducer.assume_dense = True     # every replay produces every gradient on every rank
                    self.reducer.frozen_grads = True     # ... into the tensors g_opt was captured on
                if self.static_exchange:
                    self.reducer.freeze()                # bucket / payload layout is what the graphs were captured on
                if hasattr(self.opt, "frozen_grads"):
                    self.opt.frozen_grads = True
                if dump:
                    g_fb.debug_dump(os.path.join(dump, "g_fb.dot"))
                    g_opt.debug_dump(os.path.join(dump, "g_opt.dot"))
            except RuntimeError as err:  # keep training eagerly rather than die on a capture restriction
                if self.reducer is not None:
                    # a multi-rank run where ONE rank falls back to eager steps would leave the ranks with different
                    # exchange layouts (and different collectives per step): fatal, not a warning
                    raise RuntimeError(f"HIP-graph capture failed on a data-parallel rank: {err}") from err
                import warnings
                warnings.warn(f"HIP-graph capture failed ({err}); continuing with the eager static step")
                self.use_graph = False
                torch.cuda.synchronize()
                self.model.implicit_network._force_repack = True
                self._eager_iteration()
                return self.out, self.loss_out

        self.g_fb.replay()
        self._exchange()
        self.g_opt.replay()
        # the replayed optimizer wrote the parameters behind torch's back: the packed SDF images now in memory were
        # built (inside g_fb) from the PREVIOUS values - any eager user (net.sdf, eval, plots) must re-pack
        _lib.bump_param_epoch()
        if self.sync_each_step:    # debugging aid only (see __init__)
            torch.cuda.synchronize()
        return self.out, self.loss_out
