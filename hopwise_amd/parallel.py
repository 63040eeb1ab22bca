"""One-process-per-GPU building blocks over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference only knows replicated tables under DDP (trainer/trainer.py:83-84, configurator.py:557-562).
This module adds the layout BASELINE.json's config 4 asks for: entity / user tables sharded row-wise
(owner = id % P, local row = id // P) with an all-to-all lookup, while relation tables and MLPs stay replicated.

One lookup (forward), with no host read anywhere on the way, in TWO collectives:

    hw_shard_plan (csrc/shard.hip)     ids -> exact dedup + bucket by owner, on the device: per-owner counts, the
                                       distinct local rows of each owner (ascending), inv[i] = owner * T * cap + slot
    all_to_all_single                  ONE int32 message per peer: [count, local rows...] = [P, 1 + cap], fixed size
    hw_shard_gather x T                the owner copies the requested rows of its T tables (re, im, ...) into [P, T, cap, d]
    all_to_all_single                  ONE float message per peer with every table's rows; the receive buffer holds the
                                       T compact tables (table t starts t * cap rows in; overlapping views, no copy)

`cap` is a fixed per-owner capacity (ceil(n / P * capacity_factor) + 64, at most min(n, rows per shard)) with n = the
lookup's `max_ids` -- a number fixed per kind of lookup and IDENTICAL ON EVERY RANK (RowShardedKGE freezes it at the first
batch of a kind as the maximum over ranks, or takes it from its `max_ids` argument): equal-split all-to-alls need equal
message sizes on all ranks, and a ragged last batch must not change them.  Message sizes therefore never depend on data and
nothing is read back on the good path.  An owner asked for more than `cap` distinct rows raises a device flag on the
REQUESTING rank; RowShardedKGE max-reduces that flag over the ranks right after a step's lookups (one 4-byte all-reduce), so
the decision is the same everywhere: the step's loss comes back NaN on EVERY rank (decided on the device: the reference
trainer's `_check_nan` stops all of them before backward, trainer.py:337-339 -- a rank that raised alone would leave its
peers waiting in the backward all-to-all), `sync_grads` checks the flag before any optimizer step can apply the clamped rows,
and the evaluation entry points check it before returning.  A batch LARGER than `max_ids` takes the same road: it is routed
like any other, and either fits the capacity or raises the flag for everybody.
The fused scorers then run unchanged on the compact tables with remapped ids, so the kernels never see the
sharding.  Backward is the mirror image: compact row-grads --all_to_all--> hw_shard_scatter_add into the owner's
gradient shard.  xGMI is point-to-point (7 links per GPU): all-to-all uses all links at once; there is no ring
all-reduce on the big tables at all -- only the small replicated parameters are all-reduced.

``set_shard_ops`` replaces the three device primitives -- used ONLY by the gloo/CPU tests of the exchange logic.
"""

from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from . import ops

_shard_ops = {"plan": None, "gather": None, "scatter_add": None, "full_sort": None, "full_sort_topk": None, "topk": None}


def set_shard_ops(**fns):
    """Override device primitives (plan, gather, scatter_add, full_sort, full_sort_topk, topk) with CPU stand-ins -- tests of
    the exchange logic under gloo only.  No arguments restores the HIP kernels."""
    for k in _shard_ops:
        _shard_ops[k] = fns.pop(k, None)
    if fns:
        raise KeyError(f"unknown shard ops: {sorted(fns)}")


def _exchange(P):
    """Whether the collectives run: always for P > 1; HW_FORCE_COLLECTIVES=1 also runs them in a one-rank group (a test
    hook: it puts the RCCL all_to_all / all_gather calls on the stream of a single-GPU box)."""
    import os

    return P > 1 or (os.environ.get("HW_FORCE_COLLECTIVES", "0") == "1" and dist.is_available() and dist.is_initialized())


def _staged(group, *tensors):
    """True when the collective must be staged through host memory: a non-RCCL backend (gloo: the rehearsal backend of
    the tests and of bench.py's HW_BENCH_BACKEND=gloo) asked to move device tensors.  The production path is RCCL on the
    device buffers; this branch exists so that the multi-rank exchange can run end to end with the real kernels on a box
    whose ranks share one GPU."""
    return dist.get_backend(group) != "nccl" and any(t.is_cuda for t in tensors)


def all_to_all(out, inp, group=None):
    if _staged(group, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, group=group)
    return out


def all_reduce_sum(t, group=None):
    if _staged(group, t):
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)
    return t


def all_reduce_max(t, group=None):
    if _staged(group, t):
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def all_gather(parts, t, group=None):
    if _staged(group, t):
        cp = [torch.empty(p.shape, dtype=p.dtype) for p in parts]
        dist.all_gather(cp, t.cpu(), group=group)
        for p, c in zip(parts, cp):
            p.copy_(c)
    else:
        dist.all_gather(parts, t, group=group)
    return parts


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_rows(full, rank, P):
    """Rows rank, rank + P, ... of a [N, d] table (owner = id % P, local row = id // P)."""
    return full[rank::P].contiguous()


def local_rows(n_rows, rank, P):
    return (n_rows - rank + P - 1) // P


def capacity(n_ids, n_rows, P, factor):
    """Distinct rows one owner can be asked for in one lookup of ``n_ids`` ids."""
    worst = max(1, min(int(n_ids), (int(n_rows) + P - 1) // P))
    if P == 1:
        return worst
    return max(1, min(worst, int(n_ids / P * factor) + 64))


class _Plan:
    """Routing of one lookup, shared by every table looked up with the same ids.  Device-resident; no host sync.
    One int32 message per peer carries the count and the local rows ([P, 1 + cap])."""

    def __init__(self, ids, n_rows, group, factor, overflow, n_tables, max_ids=None):
        rank, P = world(group)
        self.group, self.P, self.T = group, P, int(n_tables)
        # the capacity comes from max_ids alone, never from this rank's batch: a rank must not decide by itself to refuse (its
        # peers would wait in the collectives below).  More ids than max_ids are routed like any others; if an owner's share
        # does not fit, the overflow flag says so -- to every rank, once the caller has reduced it (RowShardedKGE does)
        n_cap = ids.numel() if max_ids is None else int(max_ids)
        self.cap = capacity(n_cap, n_rows, P, factor)
        plan = _shard_ops["plan"] or ops.shard_plan
        self.inv, self.send_meta = plan(ids, n_rows, P, self.cap, overflow, self.T)
        if _exchange(P):
            self.recv_meta = torch.empty_like(self.send_meta)
            all_to_all(self.recv_meta, self.send_meta, group)
        else:
            self.recv_meta = self.send_meta
        self.rows = P * self.T * self.cap  # rows of the packed row message


def _table_views(buf, T, cap, d):
    """The T compact tables inside one [P, T, cap, d] message: table t starts t * cap rows in and is indexed by
    inv = owner * (T * cap) + slot -- overlapping contiguous [rows, d] views, no copy."""
    flat = buf.view(-1)
    rows = buf.shape[0] - (T - 1) * cap  # the same row count for every table (the scorers ask for that); the largest
    return tuple(flat[t * cap * d : (t * cap + rows) * d].view(rows, d) for t in range(T))  # index is rows - 1


class _ShardedLookup(torch.autograd.Function):
    """All tables of a lookup travel in ONE row message per direction: [P, T, cap, d] (xGMI wants few, large collectives)."""

    @staticmethod
    def forward(ctx, plan, *shards):
        gather = _shard_ops["gather"] or ops.shard_gather
        d = shards[0].shape[1]
        send = torch.empty((plan.rows, d), dtype=shards[0].dtype, device=shards[0].device)
        for t, s in enumerate(shards):
            gather(s, plan.recv_meta, plan.cap, send, t, plan.T)
        if _exchange(plan.P):
            back = torch.empty_like(send)
            all_to_all(back, send, plan.group)
        else:
            back = send
        ctx.plan = plan
        ctx.save_for_backward(*shards)
        return _table_views(back, plan.T, plan.cap, d)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        shards = ctx.saved_tensors
        scatter = _shard_ops["scatter_add"] or ops.shard_scatter_add_
        d, P, T, cap = shards[0].shape[1], plan.P, plan.T, plan.cap
        if not _exchange(P) and P == 1:  # nothing travels: table t's gradient rows [0, cap) are the owner's rows already
            outs = []
            for s, g in zip(shards, grads):
                sink = ops._live_sink(s) if not torch.is_grad_enabled() else None
                dst = sink[0] if sink is not None else torch.zeros_like(s)
                scatter(dst, plan.recv_meta, cap, g.contiguous()[:cap], 0, 1)
                if sink is not None:
                    sink[1].touch(s.data_ptr())
                outs.append(None if sink is not None else dst)
            return (None, *outs)
        if T == 1:
            send = grads[0].contiguous()
        else:  # pack the T dense compact gradients into the message layout (chunk (p, t) = rows p * T * cap .. + cap of g_t)
            send = torch.empty((plan.rows, d), dtype=grads[0].dtype, device=grads[0].device)
            sv = send.view(P, T, cap, d)
            for t, g in enumerate(grads):
                sv[:, t].copy_(g.contiguous().as_strided((P, cap, d), (T * cap * d, d, 1)))
        if _exchange(P):
            recv = torch.empty_like(send)
            all_to_all(recv, send, plan.group)
        else:
            recv = send
        outs = []
        for t, s in enumerate(shards):
            sink = ops._live_sink(s) if not torch.is_grad_enabled() else None
            if sink is not None:  # optim.FusedAdam's persistent .grad: add the rows there, nothing for autograd to add
                scatter(sink[0], plan.recv_meta, cap, recv, t, T)
                sink[1].touch(s.data_ptr())
                outs.append(None)
                continue
            gs = torch.zeros_like(s)
            scatter(gs, plan.recv_meta, cap, recv, t, T)
            outs.append(gs)
        return (None, *outs)


def sharded_lookup(shards, ids, n_rows, group=None, capacity_factor=1.25, overflow=None, max_ids=None):
    """Rows of the row-sharded tables ``shards`` (same sharding, e.g. re/im; ``n_rows`` = rows of the full table) for
    ``ids``.  Returns (compact tables tuple (views of one [P * T * cap, d] buffer), inverse map shaped like ids, overflow
    flag tensor).  ``max_ids``: the id count the exchange capacity is sized from -- it MUST be the same on every rank
    (default: this call's ids.numel(), i.e. callers guarantee equal batch sizes); fewer ids than that are fine, more are
    routed too and raise the flag if an owner's share exceeds the capacity.  The flag is LOCAL to the requesting rank: a
    caller that acts on it must max-reduce it over the group first (``all_reduce_max``), or the ranks will disagree."""
    if overflow is None:
        overflow = torch.zeros(1, dtype=torch.int32, device=ids.device)
    plan = _Plan(ids, n_rows, group, capacity_factor, overflow, len(shards), max_ids)
    return _ShardedLookup.apply(plan, *shards), plan.inv, overflow


class _GradScale(torch.autograd.Function):
    """Identity whose backward multiplies by ``scale`` (1 / P: ranks' losses are averaged, like DDP averages gradients)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


class RowShardedKGE(nn.Module):
    """A KGE model (hopwise_amd.model.{TransE,RotatE,ComplEx,DistMult}) with its user and entity tables
    row-sharded over the process group.  ``calculate_loss`` has the reference semantics on this rank's batch (its
    backward is pre-scaled by 1 / P); call :meth:`sync_grads` before the optimizer step (only the replicated relation
    tables are all-reduced -- the big tables' row gradients went back to their owners in the reverse all-to-all).

    Evaluation shards the candidate tails instead: every rank scores the same queries
    against its own rows, keeps a local top-k, and ``merge_topk`` ranks the k * P candidates."""

    def __init__(self, model, group=None, capacity_factor=1.25, max_ids=None):
        """``max_ids``: optional {"train_entity": n, "train_user": n, "eval_rec": n, "eval_kg": n} -- the largest id count
        of each kind of lookup, identical on every rank (e.g. 5 * train_batch_size entity ids per training step).  Kinds
        not given are frozen at their first batch as the maximum over ranks (one all-reduce, once); later batches may be
        smaller (a ragged last batch) but not larger."""
        super().__init__()
        self._max_ids = dict(max_ids or {})
        self.group = group
        self.rank, self.P = world(group)
        self.model = model
        self.capacity_factor = float(capacity_factor)
        self.n_users, self.n_entities, self.n_items = model.n_users, model.n_entities, getattr(model, "n_items", None)
        for n in model.USER_TABLES + model.ENTITY_TABLES:
            emb = getattr(model, n)
            sh = nn.Parameter(shard_rows(emb.weight.data, self.rank, self.P))
            delattr(model, n)
            self.register_parameter("shard_" + n, sh)
        self._compact = None
        self._overflow = None  # int32 [1] on the tables' device: max distinct rows asked of one owner, when > cap
        self._ovf_host, self._ovf_event = None, None
        outer = self

        def _tables(side):
            c = outer._compact
            head = c["user"] if side == "rec" else c["entity"]
            return head, model._w(model.RELATION_TABLES), c["entity"]

        model._tables = _tables
        model._narrow_rec_tails = False  # compact tables are indexed by remapped ids, not by entity id

    def _shards(self, names):
        return tuple(getattr(self, "shard_" + n) for n in names)

    # -- overflow of the fixed exchange capacity: detected on the device, reported one call late, never a stall --------
    def _flag(self, device):
        if self._overflow is None or self._overflow.device != device:
            self._overflow = torch.zeros(1, dtype=torch.int32, device=device)
        return self._overflow

    def check_overflow(self, wait=True):
        """Raise if a lookup asked one owner for more distinct rows than the exchange capacity.
        wait=False: only what has already landed in the pinned copy (no stall at all);
        wait="lookups": wait for the pinned copy posted right after the LAST lookups (the routing kernels run at the very
        start of a step, so the host waits for those few microseconds of device work, not for the step);
        wait=True: additionally read the device flag itself (a full stream sync: tests, end of a run)."""
        if self._overflow is None:
            return
        if self._overflow.is_cuda:
            if self._ovf_event is None:
                if not wait:
                    return
                seen = int(self._overflow.item())
            else:
                if wait or self._ovf_event.query():
                    self._ovf_event.synchronize()
                    seen = int(self._ovf_host.item())
                else:
                    seen = 0
                if wait is True:
                    seen = max(seen, int(self._overflow.item()))
        else:
            seen = int(self._overflow.item())
        if seen:
            self._overflow.zero_()  # reported once: the caller may raise capacity_factor / max_ids and go on
            if self._ovf_host is not None:
                self._ovf_host.zero_()
            raise RuntimeError(
                f"row-sharded lookup: an owner was asked for {seen} distinct rows by one rank, more than the exchange "
                f"capacity (capacity_factor={self.capacity_factor}, max_ids={self._max_ids}); the affected step is invalid on "
                f"every rank -- raise capacity_factor (P = {self.P} is the worst case) or max_ids")

    def _post_overflow_copy(self):
        """After the LAST lookup of a call: make the flag the maximum over ranks (a 4-byte all-reduce; every rank calls this at
        the same point of the same call, so it lines up like the all-to-alls around it), then post the pinned copy."""
        if self._overflow is not None and _exchange(self.P):
            all_reduce_max(self._overflow, self.group)
        if self._overflow is not None and self._overflow.is_cuda:
            if self._ovf_host is None:
                self._ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._ovf_event = torch.cuda.Event()
            self._ovf_host.copy_(self._overflow, non_blocking=True)
            self._ovf_event.record()

    def _frozen_max_ids(self, kind, n):
        """The id count lookups of ``kind`` size their messages from: given at construction, or frozen at the first batch
        as the maximum over ranks (ranks may see different first batches; the all-to-alls need one size)."""
        m = self._max_ids.get(kind)
        if m is None:
            m = int(n)
            if _exchange(self.P):
                t = torch.tensor([m], dtype=torch.int64)
                if dist.get_backend(self.group) == "nccl":  # RCCL reduces device tensors
                    t = t.to(self._shards(self.model.ENTITY_TABLES)[0].device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                m = int(t.item())
            self._max_ids[kind] = m
        return m

    def _lookup(self, names, ids, n_rows, kind):
        return sharded_lookup(self._shards(names), ids, n_rows, self.group, self.capacity_factor, self._flag(ids.device),
                              self._frozen_max_ids(kind, ids.numel()))[:2]

    def calculate_loss(self, interaction):
        m = self.model
        # the previous call's flag (already the maximum over ranks).  At P > 1 the check must come out the same on every rank,
        # so it waits for that call's pinned copy (posted right after its routing kernels: landed long ago) instead of
        # polling it -- a poll could be seen by one rank and missed by another, which would then wait in the all-to-all
        self.check_overflow(wait="lookups" if _exchange(self.P) else False)
        keys = (m.ITEM_ID, m.NEG_ITEM_ID, m.HEAD_ENTITY_ID, m.TAIL_ENTITY_ID, m.NEG_TAIL_ENTITY_ID)
        ent_ids = torch.cat([interaction[k].reshape(-1) for k in keys])
        ent_rows, ent_inv = self._lookup(m.ENTITY_TABLES, ent_ids, self.n_entities, "train_entity")
        usr_rows, usr_inv = self._lookup(m.USER_TABLES, interaction[m.USER_ID], self.n_users, "train_user")
        self._post_overflow_copy()
        pi, ni, h, t, nt = torch.split(ent_inv, [interaction[k].numel() for k in keys])
        remapped = {m.USER_ID: usr_inv, m.ITEM_ID: pi, m.NEG_ITEM_ID: ni, m.HEAD_ENTITY_ID: h,
                    m.RELATION_ID: interaction[m.RELATION_ID], m.TAIL_ENTITY_ID: t, m.NEG_TAIL_ENTITY_ID: nt}
        self._compact = {"user": usr_rows, "entity": ent_rows}
        try:
            losses = m.calculate_loss(remapped)
        finally:
            self._compact = None
        loss = sum(losses) if isinstance(losses, tuple) else losses
        # an overflowed lookup clamped ids into the compact tables: this step's loss is NaN (decided on the device, no stall),
        # which the trainer's NaN check turns into an error before backward (trainer.py:337-339)
        consts = self.__dict__.get("_nan_consts")
        if consts is None or consts[0].device != loss.device:
            consts = self.__dict__["_nan_consts"] = (torch.full((1,), float("nan"), device=loss.device, dtype=loss.dtype),
                                                     torch.zeros(1, device=loss.device, dtype=loss.dtype))
        loss = loss + torch.where(self._overflow.to(loss.device) > 0, consts[0], consts[1])[0]
        return _GradScale.apply(loss, 1.0 / self.P) if self.P > 1 else loss

    def sync_grads(self):
        """Sum the replicated parameters' gradients over ranks (each already carries the 1 / P of the loss average): ONE
        all-reduce of the flattened gradients (xGMI wants few, large collectives).  Checks the overflow flag first -- this
        runs between backward and the optimizer step, so a step whose lookups were clamped is never applied (the wait is for
        the pinned copy posted right after this step's routing kernels, not for the step)."""
        self.check_overflow(wait="lookups")
        if not _exchange(self.P):
            return
        grads = [p.grad for n, p in self.named_parameters() if p.grad is not None and not n.startswith("shard_")]
        if not grads:
            return
        if len(grads) == 1:
            all_reduce_sum(grads[0], self.group)
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        all_reduce_sum(flat, self.group)
        off = 0
        for g in grads:
            g.copy_(flat[off : off + g.numel()].view_as(g))
            off += g.numel()

    # -- evaluation: tails sharded, queries replicated (every rank passes the SAME interaction) ----------------------------
    def _queries(self, interaction, kg):
        m = self.model
        if kg:
            rows, inv = self._lookup(m.ENTITY_TABLES, interaction[m.HEAD_ENTITY_ID], self.n_entities, "eval_kg")
            self._post_overflow_copy()
            return rows, inv, interaction[m.RELATION_ID], 0
        rows, inv = self._lookup(m.USER_TABLES, interaction[m.USER_ID], self.n_users, "eval_rec")
        self._post_overflow_copy()
        return rows, inv, None, m._rec_rel_full_sort()

    def _local_cols(self, kg):
        n = self.n_entities if kg else self.n_items
        return n, local_rows(n, self.rank, self.P)

    def _gemm_eval(self, n_queries):
        """DistMult / ComplEx / Analogy: this rank's [B, n_loc] block of the full sort is one product of the query rows with its
        own tail shard on the MFMA GEMM (the model's ``gemm_full_sort`` switch); the device primitives replaced for the CPU
        tests keep the gather form."""
        m = self.model
        return (getattr(m, "BILINEAR", False) and m.gemm_full_sort and n_queries > 0
                and _shard_ops["full_sort"] is None and _shard_ops["full_sort_topk"] is None)

    def _local_scores_gemm(self, head, inv, r_idx, r_fixed, n_loc):
        m = self.model
        shards = self._shards(m.ENTITY_TABLES)
        q, _ = ops.kge_query_rows(m.MODEL, head, m._w(m.RELATION_TABLES), shards, inv, r_idx, r_fixed)
        if n_loc == 0:
            return torch.empty((q.shape[0], 0), dtype=torch.float32, device=q.device)
        return ops.gemm(q, m._bilinear_table(shards, n_loc))

    @torch.no_grad()
    def full_sort_topk(self, interaction, k, history_index=None, kg=False):
        """Global masked top-k (trainer.py:731-734 + collector.py:173-183): (values [B, k], global ids [B, k]), ordered
        (value desc, id asc) exactly like the unsharded ``full_sort_topk``.  Rec side: candidate items are the entity rows
        [0, n_items), of which this rank owns ids rank, rank + P, ..."""
        m = self.model
        head, inv, r_idx, r_fixed = self._queries(interaction, kg)
        n, n_loc = self._local_cols(kg)
        hist = None
        if history_index is not None:  # entries owned elsewhere get column -1, which the mask builder ignores
            hr, hc = history_index
            hist = (hr, torch.where(hc % self.P == self.rank, torch.div(hc, self.P, rounding_mode="floor"),
                                    torch.full_like(hc, -1)))
        kk = min(int(k), n_loc)
        if self._gemm_eval(inv.numel()):
            local = self._local_scores_gemm(head, inv, r_idx, r_fixed, n_loc)
            ops.mask_scores_(local, hist, column0=self.rank == 0)
            val, idx = ops.topk_rows(local, kk)
        else:
            val, idx = (_shard_ops["full_sort_topk"] or ops.kge_full_sort_topk)(
                m.MODEL, head, m._w(m.RELATION_TABLES), self._shards(m.ENTITY_TABLES), inv, r_idx, r_fixed, 0, n_loc, kk, hist,
                self.rank == 0, m.margin)
        gid = idx * self.P + self.rank  # column 0 = [PAD] lives on rank 0 only
        if kk < k:  # fewer local candidates than k: pad with -inf / ids past the table so they lose every comparison
            pad = int(k) - kk
            val = torch.cat([val, torch.full((val.shape[0], pad), float("-inf"), device=val.device)], 1)
            gid = torch.cat([gid, torch.full((gid.shape[0], pad), n, dtype=gid.dtype, device=gid.device)], 1)
        out = merge_topk(val, gid, int(k), self.group)
        self.check_overflow(wait="lookups")  # an overflowed query lookup scored the wrong rows: never return those
        return out

    @torch.no_grad()
    def _full_sort(self, interaction, kg):
        m = self.model
        head, inv, r_idx, r_fixed = self._queries(interaction, kg)
        n, n_loc = self._local_cols(kg)
        if self._gemm_eval(inv.numel()):
            local = self._local_scores_gemm(head, inv, r_idx, r_fixed, n_loc)
        else:
            local = (_shard_ops["full_sort"] or ops.kge_full_sort)(
                m.MODEL, head, m._w(m.RELATION_TABLES), self._shards(m.ENTITY_TABLES), inv, r_idx, r_fixed, 0, n_loc,
                m.margin).view(-1, n_loc)
        self.check_overflow(wait="lookups")
        if not _exchange(self.P):
            return local
        B, width = local.shape[0], local_rows(n, 0, self.P)
        pad = torch.zeros((B, width), dtype=local.dtype, device=local.device)
        pad[:, :n_loc] = local
        parts = [torch.empty_like(pad) for _ in range(self.P)]
        all_gather(parts, pad, self.group)
        full = torch.stack(parts, dim=2).reshape(B, width * self.P)[:, :n]  # column = local * P + rank
        return full.contiguous()

    def full_sort_predict(self, interaction):
        """[B, n_items] (trainer.py:731 views it as that); all ranks pass the same interaction."""
        return self._full_sort(interaction, kg=False)

    def full_sort_predict_kg(self, interaction):
        return self._full_sort(interaction, kg=True)

    def full_state_dict(self):
        """Reassemble the reference-compatible state_dict (all ranks participate; rank 0 gets full tables)."""
        out = {k: v for k, v in self.model.state_dict().items()}
        for n in self.model.USER_TABLES + self.model.ENTITY_TABLES:
            sh = getattr(self, "shard_" + n).data
            n_full = self.n_users if n in self.model.USER_TABLES else self.n_entities
            if self.P == 1:
                out[n + ".weight"] = sh.clone()
                continue
            pad = torch.zeros((local_rows(n_full, 0, self.P), sh.shape[1]), dtype=sh.dtype, device=sh.device)
            pad[: sh.shape[0]] = sh
            parts = [torch.empty_like(pad) for _ in range(self.P)]
            all_gather(parts, pad, self.group)
            full = torch.empty((n_full, sh.shape[1]), dtype=sh.dtype, device=sh.device)
            for r in range(self.P):
                full[r :: self.P] = parts[r][: local_rows(n_full, r, self.P)]
            out[n + ".weight"] = full
        return out


def merge_topk(local_vals, local_ids, k, group=None, topk_fn=None):
    """Global top-k from per-rank top-k candidates: all_gather k*P (value, global id) pairs, then rank them
    (value desc, id asc).  ``topk_fn(values [B, kP], k)`` defaults to the HIP row top-k."""
    rank, P = world(group)
    if not _exchange(P):
        return local_vals[:, :k], local_ids[:, :k]
    vals = [torch.empty_like(local_vals) for _ in range(P)]
    ids = [torch.empty_like(local_ids) for _ in range(P)]
    all_gather(vals, local_vals.contiguous(), group)
    all_gather(ids, local_ids.contiguous(), group)
    v, i = torch.cat(vals, dim=1), torch.cat(ids, dim=1)
    # order candidates by global id first so that "index asc" among equal values means "global id asc"
    order = torch.argsort(i, dim=1, stable=True)
    v, i = torch.gather(v, 1, order), torch.gather(i, 1, order)
    tv, ti = (topk_fn or _shard_ops["topk"] or ops.topk_rows)(v.contiguous(), k)
    return tv, torch.gather(i, 1, ti.clamp_min(0))


# ---- replicas: the PGPR / CAFE / TPRec searches (graph, tables and policy replicated, users partitioned,
# no collective inside a search; only the per-user results are gathered for the metrics) --------------------------------------
def replica_users(users, group=None):
    """This rank's share of an evaluation batch of users: positions rank, rank + P, ... (round-robin, so every rank gets the
    same count +- 1 whatever the order of `users`)."""
    rank, P = world(group)
    return users[rank::P]


def gather_user_results(local, n_total, group=None, fill=0):
    """Inverse of replica_users for per-user result rows: `local` [ceil-or-floor(n_total / P), ...] on every rank -> the
    [n_total, ...] tensor in the original user order, on every rank.  Ranks with one row fewer are padded for the
    fixed-size all_gather and the padding is dropped."""
    rank, P = world(group)
    if P == 1:
        return local
    per = (n_total + P - 1) // P
    pad = local
    if local.shape[0] < per:
        pad = torch.cat([local, torch.full((per - local.shape[0],) + tuple(local.shape[1:]), fill, dtype=local.dtype,
                                           device=local.device)])
    parts = [torch.empty_like(pad) for _ in range(P)]
    all_gather(parts, pad.contiguous(), group)
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(P):
        n_r = len(range(r, n_total, P))
        out[r::P] = parts[r][:n_r]
    return out


def max_over_ranks(seconds, device, group=None):
    rank, P = world(group)
    if P == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
