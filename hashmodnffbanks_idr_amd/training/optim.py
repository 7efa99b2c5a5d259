"""clip_grad_norm_ + Adam as three HIP launches (csrc/hm_optim.hip) behind the torch.optim interface.

Reference iteration tail (training/idr_train.py:128,306-309):
    torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=1.0);  self.optimizer.step()
with ``torch.optim.Adam(model.parameters(), lr)``.  ``ClipAdam.step()`` does both (``max_norm=None`` turns the
clipping off and leaves plain Adam); hyper-parameters, update formulas and the ``state_dict`` layout
(``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) are torch.optim.Adam's, so optimizer checkpoints
(``OptimizerParameters/*.pth``, idr_train.py:189-194) load both ways.  Everything stays on the device: the step
counter is a device int64, so ``step()`` can sit inside a captured HIP graph.
"""
import ctypes as C

import torch

from .. import _lib
from .._lib import check, lib, require_gpu


class ClipAdam(torch.optim.Optimizer):
    fused_clip = True   # GraphedTrainStep / parallel.train_step: do not call clip_grad_norm_ separately

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("ClipAdam: invalid hyper-parameter")
        # the group carries every key torch.optim.Adam's own group has (at their Adam defaults), so an
        # OptimizerParameters/*.pth written from here is continued by the reference's torch.optim.Adam
        # (idr_train.py:128,151-156: load_state_dict, then step() reads group['weight_decay'] etc.)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False,
                                      fused=None, decoupled_weight_decay=False))
        self.max_norm = max_norm
        self.frozen_grads = False  # GraphedTrainStep sets it after capture: .grad tensors must keep their addresses
        # parameter -> tensor holding its gradient instead of .grad (parallel.StaticGradExchange: views of the flat,
        # all-reduced bucket - the averaged gradients are read where the collective left them, nothing is copied back)
        self.grad_override = {}
        self._dev_state = {}     # device -> (per-parameter step counters int64[n_params], scratch float[2])
        self.last_grad_norm = None   # device scalar: total gradient norm before clipping (of the last step)

    def _device_state(self, dev):
        st = self._dev_state.get(dev)
        if st is None:
            n = sum(len(g["params"]) for g in self.param_groups)
            # scratch: [norm^2, norm] + one partial sum per 8192 gradient elements (hm_adam_scratch_floats), sized for
            # ALL parameters once so that its address never changes (the step may sit in a captured graph)
            n_scratch = 2 + sum((p.numel() + 8191) // 8192 for g in self.param_groups for p in g["params"])
            st = (torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n_scratch, dtype=torch.float32, device=dev))
            self._dev_state[dev] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            # the global gradient norm spans all groups; one hyper-parameter set keeps it a single pass
            raise NotImplementedError("ClipAdam supports one parameter group (as the reference runner uses)")
        grp = self.param_groups[0]
        if grp.get("weight_decay", 0) or grp.get("amsgrad", False) or grp.get("maximize", False):
            raise NotImplementedError("ClipAdam implements plain Adam (weight_decay = 0, no amsgrad / maximize), the "
                                      "reference runner's optimizer")
        ovr = self.grad_override
        plist = [(k, p) for k, p in enumerate(grp["params"]) if (p.grad is not None or p in ovr)]
        if not plist:
            return loss
        dev = plist[0][1].device
        steps, scratch = self._device_state(dev)
        table = (_lib.AdamTensor * len(plist))()
        for i, (k, p) in enumerate(plist):
            grad = ovr.get(p)
            if grad is not None:
                if grad.shape != p.shape or not grad.is_contiguous():
                    raise ValueError("ClipAdam: grad_override tensors must be contiguous and shaped like their parameter")
            else:
                grad = p.grad
            require_gpu(p, grad)
            if p.dtype != torch.float32 or grad.dtype != torch.float32 or p.device != dev:
                raise TypeError("ClipAdam: fp32 parameters on one device only")
            if not p.is_contiguous():
                raise ValueError("ClipAdam: parameters must be contiguous")
            if not grad.is_contiguous():
                if self.frozen_grads:
                    raise RuntimeError("ClipAdam: a gradient became non-contiguous after the step was captured")
                grad = p.grad = p.grad.contiguous()
            st = self.state[p]
            if not st:
                st["step"] = None    # materialised by state_dict(); the live counters sit in one device array
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            table[i] = _lib.AdamTensor(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(),
                                       st["exp_avg_sq"].data_ptr(), steps.data_ptr() + 8 * k, p.numel())
        b1, b2 = grp["betas"]
        need = check(lib().hm_adam_scratch_floats(C.cast(table, C.c_void_p), len(plist)))
        if need > scratch.numel():
            raise RuntimeError("ClipAdam: scratch buffer too small (parameter list changed after the first step?)")
        check(lib().hm_adam_step(C.cast(table, C.c_void_p), len(plist), float(grp["lr"]), float(b1), float(b2),
                                 float(grp["eps"]), float(self.max_norm) if self.max_norm else 0.0,
                                 C.c_void_p(scratch.data_ptr()), _lib.stream_ptr(plist[0][1])))
        self.last_grad_norm = scratch[1]
        _lib.bump_param_epoch()   # parameters were written through raw pointers: tensor._version did not move
        return loss

    # -- torch.optim.Adam-compatible checkpoints ---------------------------------------------------------
    def state_dict(self):
        for grp in self.param_groups:
            for k, p in enumerate(grp["params"]):
                st = self.state.get(p)
                if st:
                    dev_state = self._dev_state.get(p.device)
                    st["step"] = (dev_state[0][k].to(torch.float32).clone() if dev_state is not None
                                  else torch.zeros((), dtype=torch.float32))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._dev_state = {}
        for grp in self.param_groups:
            for k, p in enumerate(grp["params"]):
                st = self.state.get(p)
                if st and st.get("step") is not None:
                    steps, _ = self._device_state(p.device)
                    steps[k] = int(float(st["step"]))
