"""Dense Adam as ONE HIP launch over all tables of a model (``hw_adam_step``), with the gradient clear folded in.

The reference builds ``optim.Adam(params, lr, weight_decay)`` (trainer/trainer.py:189-190); on a GPU that is a few dozen
short launches per step (foreach ops over ~5 tables + the scalar bias-correction math), which together with the
per-table ``zero_grad`` fills is most of a training step at the reference's batch sizes (train-step
table).  ``FusedAdam`` keeps torch's update rule, state names (``step``, ``exp_avg``, ``exp_avg_sq``) and ``state_dict``
layout, but

* updates every parameter of a group in one kernel (16 tensors per launch) from a device-resident step counter, so
  the step is capturable in a HIP graph (``driver.GraphedTrainStep``);
* clears each gradient in the same pass, and keeps ``p.grad`` allocated: ``zero_grad`` after a ``step`` is free;
* registers those persistent gradients as *sinks* for the scorer's backward (``ops.KgeScoreGrouped``): the gradient
  kernel accumulates straight into ``p.grad`` instead of into a fresh zero-filled buffer that autograd then adds
  (so ``loss.backward()`` is the supported way to differentiate; ``torch.autograd.grad`` would see no table gradients).

There is no CPU path: parameters must be contiguous fp32 HIP tensors."""
import ctypes as C

import torch

from . import _lib, ops


class _SinkState:
    __slots__ = ("dirty",)

    def __init__(self):
        self.dirty = True


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_sinks=True,
                 decoupled_weight_decay=False):
        """``decoupled_weight_decay=True`` is torch.optim.AdamW's rule (the reference's ``learner: adamw``,
        trainer.py:191-192); its default weight_decay there is 1e-2, pass it explicitly."""
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdam: invalid hyper-parameter")
        # capturable: the step counter is a device tensor (what driver.GraphedTrainStep asks of an optimizer)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=True))
        self._sinks = _SinkState() if grad_sinks else None
        self._decoupled = bool(decoupled_weight_decay)
        self._launch_cache = {}
        devices = set()
        for group in self.param_groups:
            for p in group["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _lib.HopwiseAmdError("FusedAdam: parameters must be contiguous fp32 HIP tensors (no CPU path)")
                devices.add(p.device)
                if len(devices) > 1:  # one launch updates a whole group: one process per GPU, as everywhere in this engine
                    raise _lib.HopwiseAmdError("FusedAdam: all parameters must live on one device")
                if self._sinks is not None and p.requires_grad:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    ops.register_grad_sink(p, self._sinks)
                    p.register_post_accumulate_grad_hook(self._mark_dirty)

    def _mark_dirty(self, _p):
        self._sinks.dirty = True

    def zero_grad(self, set_to_none=True):
        """With gradient sinks the buffers stay allocated whatever ``set_to_none`` says, and are only written when
        something accumulated into them since the last ``step`` (which leaves them cleared)."""
        if self._sinks is None:
            return super().zero_grad(set_to_none)
        dirty = []
        for group in self.param_groups:
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                if p.grad is None:  # dropped behind our back (e.g. module.zero_grad()): put a cleared buffer back
                    p.grad = torch.zeros_like(p)
                elif self._sinks.dirty:
                    dirty.append(p.grad)
        if dirty:
            torch._foreach_zero_(dirty)
        self._sinks.dirty = False

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam checkpoints (the reference trainer's ``optimizer`` entry, trainer.py:296-304)."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise _lib.HopwiseAmdError("FusedAdam: amsgrad / maximize checkpoints are not supported")
            group["capturable"] = True
        self._launch_cache.clear()

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.int64, device=p.device)
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        elif st["step"].dtype != torch.int64 or not st["step"].is_cuda:  # state loaded from a torch.optim.Adam checkpoint
            st["step"] = st["step"].to(device=p.device, dtype=torch.int64)
        return st

    def _launches(self, group):
        """[(AdamArgs, step tensor)] for the group's parameters that have a gradient, 16 per launch; cached on
        the identity of every buffer involved."""
        ps = [p for p in group["params"] if p.grad is not None]
        states = [self._state_of(p) for p in ps]
        key = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(),
                     s["step"].data_ptr()) for p, s in zip(ps, states))
        hit = self._launch_cache.get(id(group))
        if hit is not None and hit[0] == key:
            return hit[1]
        out = []
        for lo in range(0, len(ps), _lib.ADAM_MAX_SEGS):
            chunk, cst = ps[lo : lo + _lib.ADAM_MAX_SEGS], states[lo : lo + _lib.ADAM_MAX_SEGS]
            step = cst[0]["step"]
            for p, s in zip(chunk, cst):
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.is_sparse:
                    raise _lib.HopwiseAmdError("FusedAdam: gradients must be dense contiguous fp32")
                if s["step"] is not step:  # one device counter per launch
                    if int(s["step"]) != int(step):
                        raise _lib.HopwiseAmdError("FusedAdam: parameters of one group are at different steps")
                    s["step"] = step
            args = _lib.AdamArgs()
            args.n_segs = len(chunk)
            for i, (p, s) in enumerate(zip(chunk, cst)):
                args.seg[i].p, args.seg[i].g = p.data_ptr(), p.grad.data_ptr()
                args.seg[i].m, args.seg[i].v = s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()
                args.seg[i].n = p.numel()
            out.append((args, step))
        self._launch_cache[id(group)] = (key, out)
        return out

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for args, step in self._launches(group):
                rc = lib.hw_adam_step(C.byref(args), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                      float(group["weight_decay"]), C.c_void_p(step.data_ptr()),
                                      (1 if self._sinks is not None else 0) | (2 if self._decoupled else 0),
                                      ops._stream())
                _lib.check(rc, "hw_adam_step")
        if self._sinks is not None:
            self._sinks.dirty = False
        return loss
