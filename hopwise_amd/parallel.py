"""One-process-per-GPU building blocks over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference only knows replicated tables under DDP (trainer/trainer.py:83-84, configurator.py:557-562).
This module adds the layout BASELINE.json's config 4 asks for: entity / user tables sharded row-wise
(owner = id % P) with an all-to-all lookup, while relation tables and MLPs stay replicated.

Exchange of one lookup (forward):   ids --unique--> bucket by owner --all_to_all(counts, ids)--> owner gathers
rows from its shard --all_to_all(rows)--> requester holds a COMPACT table [n_unique, d] + the inverse map.
The fused scorers then run unchanged on the compact tables with remapped ids, so the kernels never see the
sharding.  Backward is the mirror image: compact row-grads --all_to_all--> owner scatter-adds into its shard grad.
xGMI is point-to-point (7 links per GPU): all-to-all uses all links at once; there is no ring all-reduce on the
big tables at all -- only the small replicated parameters are all-reduced.

The row gather / scatter-add run on the HIP kernels (ops.gather_rows / ops.scatter_add_rows_).  They can be
replaced with ``set_row_ops`` -- used ONLY by the gloo/CPU tests of the exchange logic.
"""

from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from . import ops

_row_ops = {"gather": None, "scatter_add": None}


def set_row_ops(gather=None, scatter_add=None):
    """Override the row primitives (tests of the exchange logic on CPU/gloo).  None restores the HIP kernels."""
    _row_ops["gather"], _row_ops["scatter_add"] = gather, scatter_add


def _gather(table, idx):
    return (_row_ops["gather"] or ops.gather_rows)(table, idx)


def _scatter_add_(table, idx, src):
    return (_row_ops["scatter_add"] or ops.scatter_add_rows_)(table, idx, src)


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_rows(full, rank, P):
    """Rows rank, rank + P, ... of a [N, d] table (owner = id % P, local row = id // P)."""
    return full[rank::P].contiguous()


def local_rows(n_rows, rank, P):
    return (n_rows - rank + P - 1) // P


class _Plan:
    """Routing of one lookup, shared by every table looked up with the same ids."""

    def __init__(self, ids, group):
        rank, P = world(group)
        self.group, self.P = group, P
        uniq, inv = torch.unique(ids.reshape(-1), return_inverse=True)
        owner = uniq % P
        order = torch.argsort(owner, stable=True)
        self.uniq_sorted = uniq[order]
        self.order = order
        self.inv = inv.reshape(ids.shape)
        self.n_unique = uniq.numel()
        send = torch.bincount(owner, minlength=P)
        if P > 1:
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=group)
            self.send_counts, self.recv_counts = send.tolist(), recv.tolist()
            self.recv_ids = torch.empty(sum(self.recv_counts), dtype=ids.dtype, device=ids.device)
            dist.all_to_all_single(self.recv_ids, self.uniq_sorted, self.recv_counts, self.send_counts, group=group)
        else:
            self.send_counts = self.recv_counts = [self.n_unique]
            self.recv_ids = self.uniq_sorted
        self.recv_local = torch.div(self.recv_ids, P, rounding_mode="floor")


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, *shards):
        d_each = [s.shape[1] for s in shards]
        rows = torch.cat([_gather(s, plan.recv_local) for s in shards], dim=1) if len(shards) > 1 else _gather(
            shards[0], plan.recv_local)
        if plan.P > 1:
            back = torch.empty((plan.n_unique, rows.shape[1]), dtype=rows.dtype, device=rows.device)
            dist.all_to_all_single(back, rows.contiguous(), plan.send_counts, plan.recv_counts, group=plan.group)
        else:
            back = rows
        compact = torch.empty_like(back)
        compact[plan.order] = back  # undo the owner sort: row i belongs to the i-th unique id
        ctx.plan, ctx.d_each = plan, d_each
        ctx.shapes = [s.shape for s in shards]
        ctx.save_for_backward(*shards)
        return tuple(c.contiguous() for c in compact.split(d_each, dim=1)) if len(shards) > 1 else (compact,)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        shards = ctx.saved_tensors
        g = torch.cat([gi.contiguous() for gi in grads], dim=1) if len(grads) > 1 else grads[0].contiguous()
        g_sorted = g[plan.order].contiguous()
        if plan.P > 1:
            recv = torch.empty((len(plan.recv_local), g.shape[1]), dtype=g.dtype, device=g.device)
            dist.all_to_all_single(recv, g_sorted, plan.recv_counts, plan.send_counts, group=plan.group)
        else:
            recv = g_sorted
        outs = []
        for s, part in zip(shards, recv.split(ctx.d_each, dim=1)):
            sink = ops._live_sink(s) if not torch.is_grad_enabled() else None
            if sink is not None:  # optim.FusedAdam's persistent .grad: add the rows there, nothing for autograd to add
                _scatter_add_(sink[0], plan.recv_local, part.contiguous())
                sink[1].dirty = True
                outs.append(None)
                continue
            gs = torch.zeros_like(s)
            _scatter_add_(gs, plan.recv_local, part.contiguous())
            outs.append(gs)
        return (None, *outs)


def sharded_lookup(shards, ids, group=None):
    """Rows of the row-sharded tables ``shards`` (same sharding, e.g. re/im) for ``ids``.
    Returns (compact tables tuple [n_unique, d], inverse map shaped like ids)."""
    plan = _Plan(ids, group)
    return _ShardedLookup.apply(plan, *shards), plan.inv


class RowShardedKGE(nn.Module):
    """A KGE model (hopwise_amd.model.{TransE,RotatE,ComplEx,DistMult}) with its user and entity tables
    row-sharded over the process group.  ``calculate_loss`` has the reference semantics on this rank's batch;
    call :meth:`sync_grads` before the optimizer step (the replicated relation tables are all-reduced,
    everything is averaged over ranks like DDP does)."""

    def __init__(self, model, group=None):
        super().__init__()
        self.group = group
        self.rank, self.P = world(group)
        self.model = model
        self.n_users, self.n_entities = model.n_users, model.n_entities
        for n in model.USER_TABLES + model.ENTITY_TABLES:
            emb = getattr(model, n)
            sh = nn.Parameter(shard_rows(emb.weight.data, self.rank, self.P))
            delattr(model, n)
            self.register_parameter("shard_" + n, sh)
        self._compact = None
        outer = self

        def _tables(side):
            c = outer._compact
            head = c["user"] if side == "rec" else c["entity"]
            return head, model._w(model.RELATION_TABLES), c["entity"]

        model._tables = _tables
        model._narrow_rec_tails = False  # compact tables are indexed by remapped ids, not by entity id

    def _shards(self, names):
        return tuple(getattr(self, "shard_" + n) for n in names)

    def calculate_loss(self, interaction):
        m = self.model
        ent_ids = torch.cat([interaction[k].reshape(-1) for k in
                             (m.ITEM_ID, m.NEG_ITEM_ID, m.HEAD_ENTITY_ID, m.TAIL_ENTITY_ID, m.NEG_TAIL_ENTITY_ID)])
        ent_rows, ent_inv = sharded_lookup(self._shards(m.ENTITY_TABLES), ent_ids, self.group)
        usr_rows, usr_inv = sharded_lookup(self._shards(m.USER_TABLES), interaction[m.USER_ID], self.group)
        sizes = [interaction[k].numel() for k in
                 (m.ITEM_ID, m.NEG_ITEM_ID, m.HEAD_ENTITY_ID, m.TAIL_ENTITY_ID, m.NEG_TAIL_ENTITY_ID)]
        pi, ni, h, t, nt = torch.split(ent_inv, sizes)
        remapped = {m.USER_ID: usr_inv, m.ITEM_ID: pi, m.NEG_ITEM_ID: ni, m.HEAD_ENTITY_ID: h,
                    m.RELATION_ID: interaction[m.RELATION_ID], m.TAIL_ENTITY_ID: t, m.NEG_TAIL_ENTITY_ID: nt}
        self._compact = {"user": usr_rows, "entity": ent_rows}
        try:
            return m.calculate_loss(remapped)
        finally:
            self._compact = None

    def sync_grads(self):
        if self.P == 1:
            return
        for n, p in self.named_parameters():
            if p.grad is None:
                continue
            if not n.startswith("shard_"):
                dist.all_reduce(p.grad, group=self.group)
            p.grad.div_(self.P)

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
            dist.all_gather(parts, pad, group=self.group)
            full = torch.empty((n_full, sh.shape[1]), dtype=sh.dtype, device=sh.device)
            for r in range(self.P):
                full[r :: self.P] = parts[r][: local_rows(n_full, r, self.P)]
            out[n + ".weight"] = full
        return out


def merge_topk(local_vals, local_ids, k, group=None, topk_fn=None):
    """Global top-k from per-rank top-k candidates: all_gather k*P (value, global id) pairs, then rank them
    (value desc, id asc).  ``topk_fn(values [B, kP], k)`` defaults to the HIP row top-k."""
    rank, P = world(group)
    if P == 1:
        return local_vals[:, :k], local_ids[:, :k]
    vals = [torch.empty_like(local_vals) for _ in range(P)]
    ids = [torch.empty_like(local_ids) for _ in range(P)]
    dist.all_gather(vals, local_vals.contiguous(), group=group)
    dist.all_gather(ids, local_ids.contiguous(), group=group)
    v, i = torch.cat(vals, dim=1), torch.cat(ids, dim=1)
    # order candidates by global id first so that "index asc" among equal values means "global id asc"
    order = torch.argsort(i, dim=1, stable=True)
    v, i = torch.gather(v, 1, order), torch.gather(i, 1, order)
    tv, ti = (topk_fn or ops.topk_rows)(v.contiguous(), k)
    return tv, torch.gather(i, 1, ti.clamp_min(0))


def max_over_ranks(seconds, device, group=None):
    rank, P = world(group)
    if P == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
