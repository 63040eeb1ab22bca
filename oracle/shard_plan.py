"""CPU restatement of the device-side routing of a row-sharded lookup (hopwise_amd/csrc/shard.hip: hw_shard_plan,
hw_shard_gather, hw_shard_scatter_add).  TEST INFRASTRUCTURE: only tests/ import this.  There is no reference
counterpart -- hopwise replicates its tables under DDP (trainer/trainer.py:83-84); the contract restated here is
SURVEY.md §8(e) row 1 (owner = id % P, bucket by owner, all-to-all, local gather) as fixed by include/hopwise_amd.h.
"""

import numpy as np


def plan(ids, n_rows, P, cap, n_tables=1):
    """-> (inv like ids, meta int32 [P, 1 + cap] (meta[o, 0] = count, meta[o, 1:] = local rows, unused slots -1),
    overflow int).  inv = owner * (n_tables * cap) + slot."""
    ids = np.asarray(ids, np.int64)
    flat = np.clip(ids.reshape(-1), 0, n_rows - 1)
    inv = np.zeros(flat.size, np.int64)
    meta = np.full((P, 1 + cap), -1, np.int32)
    overflow = 0
    for o in range(P):
        mine = np.unique(flat[flat % P == o])  # ascending ids = ascending local rows
        if mine.size > cap:
            overflow = max(overflow, int(mine.size))
        keep = mine[:cap]
        meta[o, 0] = keep.size
        meta[o, 1 : 1 + keep.size] = keep // P
        sel = flat % P == o
        slot = np.minimum(np.searchsorted(mine, flat[sel]), cap - 1)
        inv[sel] = o * (n_tables * cap) + slot
    return inv.reshape(ids.shape), meta, overflow


def gather(shard, meta, cap, out, table=0, n_tables=1):
    """out [P * n_tables * cap, d] viewed [P, n_tables, cap, d]: fills [p, table, :count_p]."""
    P = meta.shape[0]
    v = out.reshape(P, n_tables, cap, -1)
    for p in range(P):
        c = int(meta[p, 0])
        v[p, table, :c] = shard[meta[p, 1 : 1 + c]]
    return out


def scatter_add(grad_shard, meta, cap, src, table=0, n_tables=1):
    P = meta.shape[0]
    v = src.reshape(P, n_tables, cap, -1)
    for p in range(P):
        c = int(meta[p, 0])
        np.add.at(grad_shard, meta[p, 1 : 1 + c], v[p, table, :c])
    return grad_shard
