"""CPU restatement of the device-side routing of a row-sharded lookup (hopwise_amd/csrc/shard.hip: hw_shard_plan,
hw_shard_gather, hw_shard_scatter_add).  TEST INFRASTRUCTURE: only tests/ import this.  There is no reference
counterpart -- hopwise replicates its tables under DDP (trainer/trainer.py:83-84); the contract restated here is
SURVEY.md §8(e) row 1 (owner = id % P, bucket by owner, all-to-all, local gather) as fixed by include/hopwise_amd.h.
"""

import numpy as np


def plan(ids, n_rows, P, cap):
    """-> (inv like ids, send_counts int32 [P], send_local int32 [P * cap] (unused slots -1), overflow int)."""
    ids = np.asarray(ids, np.int64)
    flat = np.clip(ids.reshape(-1), 0, n_rows - 1)
    inv = np.zeros(flat.size, np.int64)
    send_counts = np.zeros(P, np.int32)
    send_local = np.full(P * cap, -1, np.int32)
    overflow = 0
    for o in range(P):
        mine = np.unique(flat[flat % P == o])  # ascending ids = ascending local rows
        if mine.size > cap:
            overflow = max(overflow, int(mine.size))
        keep = mine[:cap]
        send_counts[o] = keep.size
        send_local[o * cap : o * cap + keep.size] = keep // P
        sel = flat % P == o
        slot = np.minimum(np.searchsorted(mine, flat[sel]), cap - 1)
        inv[sel] = o * cap + slot
    return inv.reshape(ids.shape), send_counts, send_local, overflow


def gather(shard, recv_local, recv_counts, cap, fill=0.0):
    P = len(recv_counts)
    out = np.full((P * cap, shard.shape[1]), fill, shard.dtype)
    for p in range(P):
        c = int(recv_counts[p])
        out[p * cap : p * cap + c] = shard[recv_local[p * cap : p * cap + c]]
    return out


def scatter_add(grad_shard, recv_local, recv_counts, cap, src):
    for p in range(len(recv_counts)):
        c = int(recv_counts[p])
        np.add.at(grad_shard, recv_local[p * cap : p * cap + c], src[p * cap : p * cap + c])
    return grad_shard
