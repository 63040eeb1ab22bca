"""How much of the last hop's action-set construction of a PGPR beam search is repeated work?  Rows of the hop = paths; the
action set of a row depends on (user, last node) and, through the visited filter, on the few nodes already on the path.
Counts, for the bench's 256-user search: rows, rows over max_acts neighbours, distinct (user, last node) pairs among them, and
the candidate scores (sum of degrees) looked up per row against per distinct pair."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_pgpr

dev = torch.device("cuda", 0)
kg, _, m = bench_pgpr.make_model(dev)
users = torch.from_numpy(np.random.default_rng(5).integers(1, kg.n_users, 256)).to(dev)
res = m.beam_search_device(users, keep_buffers=True)
g, buf = m.graph, res.buffers
nr = res.n_rows_per_hop.cpu().numpy()
for hop in range(len(nr)):
    n = int(nr[hop])
    rec = buf["paths"][hop][:n].long()
    nn = rec[:, 1]
    col = 2 + 3 * (nn - 2).clamp_min(0)
    typ = torch.where(nn > 1, rec.gather(1, (col + 1).view(-1, 1)).view(-1), torch.zeros_like(nn))
    nid = torch.where(nn > 1, rec.gather(1, (col + 2).view(-1, 1)).view(-1), rec[:, 0])
    gid = torch.where(typ > 0, g.n_users + nid, nid)
    deg = g.ptr[gid + 1] - g.ptr[gid]
    over = deg > m.max_acts
    key = rec[:, 0] * (g.n_users + g.n_entities) + gid
    uk, inv = torch.unique(key[over], return_inverse=True)
    d_over = deg[over]
    first = torch.zeros_like(uk).scatter_(0, inv, d_over)  # degree per distinct pair
    print(f"hop {hop + 1}: rows {n}, over-full {int(over.sum())} ({100.0 * float(over.float().mean()):.1f} %), "
          f"distinct (user, node) among them {uk.numel()} ({float(over.sum()) / max(uk.numel(), 1):.2f} rows per pair); "
          f"candidates per row summed {int(d_over.sum())}, per distinct pair {int(first.sum())}; all rows: distinct pairs "
          f"{torch.unique(key).numel()} of {n}")
