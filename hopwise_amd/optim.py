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
  Sinks bypass autograd's AccumulateGrad, hence DistributedDataParallel's reducer: with torch.distributed initialised
  at world size > 1 they are REFUSED (``grad_sinks=False`` keeps DDP correct; ``grad_sinks="caller_syncs"`` is for
  callers that all-reduce ``p.grad`` themselves, e.g. ``parallel.RowShardedKGE.sync_grads``).  Parameters no backward
  touched since the last step are skipped, as torch.optim.Adam skips ``grad is None``.

There is no CPU path: parameters must be contiguous fp32 HIP tensors."""
import ctypes as C

import torch

from . import _lib, ops


class _SinkState:
    """Bookkeeping of the gradient sinks of one optimizer: whether anything accumulated since the last step (``dirty``),
    WHICH parameters did (``touched``: torch.optim.Adam skips a parameter whose ``.grad`` is None, i.e. one no backward
    reached -- its moments do not decay and weight decay is not applied; the sink buffers are never None, so that is
    tracked here), and whether writing straight into ``p.grad`` is safe under torch.distributed.

    ``used`` / ``plain``: a persistent gradient buffer only pays for a parameter some gradient kernel writes into directly
    (the embedding tables of the scorers).  For every other parameter -- Linear / BatchNorm weights, whose gradient arrives
    through autograd's AccumulateGrad -- it costs an ``add`` launch per parameter per step (AccumulateGrad adds into an existing
    ``.grad`` and takes the incoming tensor as it is when ``.grad`` is None): 30 launches of a CAFE step, 120 of a TuckER step.
    So a parameter that has been stepped without any kernel having used its buffer (``used``) becomes ``plain``: from then on
    ``zero_grad`` drops its gradient like torch's ``set_to_none`` does, and the kernels fall back to handing autograd their
    gradient should such a parameter ever reach them."""

    __slots__ = ("dirty", "touched", "caller_syncs", "used", "plain")

    def __init__(self, caller_syncs=False):
        self.dirty = True
        self.touched = set()
        self.caller_syncs = caller_syncs
        self.used = set()
        self.plain = set()

    def touch(self, data_ptr, sink=False):
        self.dirty = True
        self.touched.add(data_ptr)
        if sink:
            self.used.add(data_ptr)
            self.plain.discard(data_ptr)

    def check_distributed(self):
        """Sinks bypass autograd's AccumulateGrad, so DistributedDataParallel's reducer hooks never fire for these
        parameters and replicas would diverge silently.  Refuse unless the caller all-reduces ``p.grad`` itself."""
        if self.caller_syncs:
            return
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise _lib.HopwiseAmdError(
                "FusedAdam gradient sinks write into p.grad without passing through autograd: under torch.distributed "
                "(world size > 1) DDP would never all-reduce them.  Build the optimizer with grad_sinks=False (DDP works "
                "unchanged), or with grad_sinks='caller_syncs' if you all-reduce p.grad yourself before step() "
                "(hopwise_amd.parallel.RowShardedKGE.sync_grads does)")


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_sinks=True,
                 decoupled_weight_decay=False):
        """``decoupled_weight_decay=True`` is torch.optim.AdamW's rule (the reference's ``learner: adamw``,
        trainer.py:191-192); its default weight_decay there is 1e-2, pass it explicitly."""
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("FusedAdam: invalid hyper-parameter")
        # capturable: the step counter is a device tensor (what driver.GraphedTrainStep asks of an optimizer)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=True))
        if grad_sinks not in (True, False, "caller_syncs"):
            raise ValueError("grad_sinks: True, False or 'caller_syncs'")
        self._sinks = _SinkState(caller_syncs=(grad_sinks == "caller_syncs")) if grad_sinks else None
        if self._sinks is not None:
            self._sinks.check_distributed()
        self._decoupled = bool(decoupled_weight_decay)
        self._launch_cache = {}
        self._counter_members, self._counter_keepalive = {}, {}
        self._pinned = set()  # ids of the step counters pin_step_groups made: never merged with another group's
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

    def _patch_step_function(self):
        """torch wraps ``step`` in ``profile_hook_step`` (a record_function scope + the hook walk): ~25 us of host time per
        call, a tenth of an eager training step at the reference's batch sizes.  ``step`` below runs the registered
        pre / post hooks itself and skips the profiler scope."""
        self._zero_grad_profile_name = f"Optimizer.zero_grad#{self.__class__.__name__}.zero_grad"

    def _mark_dirty(self, p):
        self._sinks.touch(p.data_ptr())

    def mark_touched(self, params=None):
        """For code that writes into the persistent ``p.grad`` buffers by hand (not through ``backward``): declare the
        parameters (default: all) as having a gradient this step, as assigning ``p.grad`` would for torch.optim.Adam."""
        if self._sinks is None:
            return
        for group in self.param_groups:
            for p in (group["params"] if params is None else params):
                self._sinks.touch(p.data_ptr(), sink=True)  # written by hand: keep the persistent buffer
            if params is not None:
                break

    def zero_grad(self, set_to_none=True):
        """With gradient sinks the buffers stay allocated whatever ``set_to_none`` says, and are only written when
        something accumulated into them since the last ``step`` (which leaves them cleared)."""
        if self._sinks is None:
            return super().zero_grad(set_to_none)
        dirty = []
        plain = self._sinks.plain
        for group in self.param_groups:
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                if p.data_ptr() in plain:  # no kernel writes into this one: let AccumulateGrad take the gradient as it comes
                    if set_to_none:
                        p.grad = None
                    elif p.grad is not None and self._sinks.dirty:
                        dirty.append(p.grad)
                elif p.grad is None:  # dropped behind our back (e.g. module.zero_grad()): put a cleared buffer back
                    p.grad = torch.zeros_like(p)
                elif self._sinks.dirty:
                    dirty.append(p.grad)
        if dirty:
            torch._foreach_zero_(dirty)
        self._sinks.dirty = False
        self._sinks.touched.clear()

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam checkpoints (the reference trainer's ``optimizer`` entry, trainer.py:296-304)."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise _lib.HopwiseAmdError("FusedAdam: amsgrad / maximize checkpoints are not supported")
            group["capturable"] = True
        self._launch_cache.clear()
        self._counter_members.clear()
        self._counter_keepalive.clear()
        self._pinned.clear()

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.int64, device=p.device)
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        elif st["step"].dtype != torch.int64 or not st["step"].is_cuda:  # state loaded from a torch.optim.Adam checkpoint
            st["step"] = st["step"].to(device=p.device, dtype=torch.int64)
        return st

    def pin_step_groups(self, groups):
        """``groups``: lists of parameters that are always stepped together -- the parameters of one module of a model whose
        steps use different modules from batch to batch (CAFE: one relation module per metapath level).  Each list (in
        chunks of 16, what one launch updates) gets ONE device step counter now and keeps it: counters of different groups are
        never merged however equal their values, so no later step regroups them -- which a captured step relies on, its
        launches hold the counters' addresses (driver.GraphedStagedTrainStep throws a capture away when the optimizer moved one
        of its parameters to another counter; with pinned groups that never happens).  The update rule is untouched: every
        parameter's step count is what torch.optim.Adam's would be.  A group whose members have been stepped a different number
        of times already is left alone."""
        for params in groups:
            params = [p for p in params if p.requires_grad]
            for lo in range(0, len(params), _lib.ADAM_MAX_SEGS):
                chunk = params[lo : lo + _lib.ADAM_MAX_SEGS]
                states = [self._state_of(p) for p in chunk]
                if len({int(st["step"]) for st in states}) != 1:
                    continue
                ids = {id(p) for p in chunk}
                if all(id(st["step"]) in self._pinned and self._counter_members.get(id(st["step"])) == ids for st in states):
                    continue  # pinned by an earlier call
                counter = states[0]["step"].clone()
                for st in states:
                    old = self._counter_members.get(id(st["step"]))
                    if old is not None:
                        old -= ids
                    st["step"] = counter
                self._counter_members[id(counter)] = set(ids)
                self._counter_keepalive[id(counter)] = counter
                self._pinned.add(id(counter))
        self._launch_cache.clear()

    def _stepped(self, group):
        """The group's parameters this step updates: those with a gradient and, with sinks, those a backward reached since
        the last step (as torch.optim.Adam skips ``grad is None``)."""
        ps = [p for p in group["params"] if p.grad is not None]
        if self._sinks is not None:
            ps = [p for p in ps if p.data_ptr() in self._sinks.touched]
        return ps

    def _launches(self, group):
        """[(AdamArgs, step tensor, parameters)] for the group's parameters that have a gradient, 16 per launch; cached per
        composition on the identity of every persistent buffer involved (the gradient pointers are filled in per step).

        A launch reads ONE device step counter, so parameters share a counter only while they are always stepped
        together: a parameter that sits a step out (untouched, see _SinkState) is split off with its own counter."""
        ps = self._stepped(group)
        states = [self._state_of(p) for p in ps]
        key = tuple((p.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), s["step"].data_ptr())
                    for p, s in zip(ps, states))
        cache = self._launch_cache.setdefault(id(group), {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        for p in ps:
            if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.is_sparse:
                raise _lib.HopwiseAmdError("FusedAdam: gradients must be dense contiguous fp32")
        # 1. counters shared with a parameter that is NOT in this composition are split (clone keeps the value)
        present = {id(p) for p in ps}
        by_counter = {}
        for p, s in zip(ps, states):
            by_counter.setdefault(id(s["step"]), []).append((p, s))
        for cid, members in by_counter.items():
            owners = self._counter_members.get(cid)
            if owners is not None and not owners <= present:
                fresh = members[0][1]["step"].clone()
                self._pinned.discard(cid)  # a pinned group stepped in part: its counter is an ordinary one from here on
                for _, s in members:
                    s["step"] = fresh
                self._counter_members[cid] = owners - {id(p) for p, _ in members}
                self._counter_members[id(fresh)] = {id(p) for p, _ in members}
        # 2. counters at the same value that only serve parameters of this composition are merged: one counter per launch
        #    (the kernel advances the counter it is given, so the 16-segment chunks of a long list each own one)
        by_value, by_pin = {}, {}
        for p, s in zip(ps, states):
            if id(s["step"]) in self._pinned:
                by_pin.setdefault(id(s["step"]), []).append((p, s))
            else:
                by_value.setdefault(int(s["step"]), []).append((p, s))  # device read: composition changes only
        out = []
        for members in by_pin.values():  # a pinned group (whole: a partial one was split and unpinned above): one launch
            args = _lib.AdamArgs()
            args.n_segs = len(members)
            for i, (p, s) in enumerate(members):
                args.seg[i].p, args.seg[i].g = p.data_ptr(), p.grad.data_ptr()
                args.seg[i].m, args.seg[i].v = s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()
                args.seg[i].n = p.numel()
            out.append((args, members[0][1]["step"], [p for p, _ in members]))
        for _, members in sorted(by_value.items()):
            for lo in range(0, len(members), _lib.ADAM_MAX_SEGS):
                chunk = members[lo : lo + _lib.ADAM_MAX_SEGS]
                ids = {id(p) for p, _ in chunk}
                shared = None
                for _, s in chunk:  # reuse a counter that already serves exactly (a subset of) this chunk
                    if self._counter_members.get(id(s["step"]), ids) <= ids:
                        shared = s["step"]
                        break
                if shared is None:
                    shared = chunk[0][1]["step"].clone()
                for _, s in chunk:
                    if s["step"] is not shared:
                        old = self._counter_members.get(id(s["step"]))
                        if old is not None:
                            old -= ids
                        s["step"] = shared
                self._counter_members[id(shared)] = set(ids)
                self._counter_keepalive[id(shared)] = shared  # ids stay unique while the bookkeeping refers to them
                args = _lib.AdamArgs()
                args.n_segs = len(chunk)
                for i, (p, s) in enumerate(chunk):
                    args.seg[i].p, args.seg[i].g = p.data_ptr(), p.grad.data_ptr()
                    args.seg[i].m, args.seg[i].v = s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()
                    args.seg[i].n = p.numel()
                out.append((args, shared, [p for p, _ in chunk]))
        # the merge may have changed step pointers: file the result under the key the NEXT identical call will compute
        key = tuple((p.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), s["step"].data_ptr())
                    for p, s in zip(ps, states))
        cache[key] = out
        return out

    def step(self, closure=None):
        # torch's profile_hook_step wrapper is skipped (it costs ~10 us of Python per step); the hooks it would run are
        # not: the process-global optimizer step hooks first, then this instance's, as torch.optim.Optimizer orders them
        from torch.optim import optimizer as _topt

        args = (closure,) if closure is not None else ()
        g_pre = getattr(_topt, "_global_optimizer_pre_hooks", None)
        g_post = getattr(_topt, "_global_optimizer_post_hooks", None)
        for hooks in (g_pre, self._optimizer_step_pre_hooks):
            if hooks:
                for hook in list(hooks.values()):
                    hook(self, args, {})
        loss = self._step(closure)
        for hooks in (self._optimizer_step_post_hooks, g_post):
            if hooks:
                for hook in list(hooks.values()):
                    hook(self, args, {})
        return loss

    @torch.no_grad()
    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            # hw_adam_step writes the parameters through raw pointers, which autograd's version counters do not see:
            # bump them by hand, as every in-place torch op would, so that anything derived from a weight and cached on
            # (data_ptr, _version) -- CAFE's transposed module weights, PGPR's pre-split policy planes -- is rebuilt
            stepped = self._stepped(group)
            if stepped:
                torch.autograd.graph.increment_version(stepped)
            for args, step, members in self._launches(group):
                for i, p in enumerate(members):  # plain parameters get a new gradient tensor every step
                    args.seg[i].g = p.grad.data_ptr()
                rc = lib.hw_adam_step(C.byref(args), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                      float(group["weight_decay"]), C.c_void_p(step.data_ptr()),
                                      (1 if self._sinks is not None else 0) | (2 if self._decoupled else 0),
                                      ops._stream())
                _lib.check(rc, "hw_adam_step")
        if self._sinks is not None:
            # a parameter stepped without a kernel having written into its buffer gets no persistent buffer from here on
            self._sinks.plain.update(self._sinks.touched - self._sinks.used)
            self._sinks.dirty = False
            self._sinks.touched.clear()
        return loss
