#!/usr/bin/env python
"""PGPR.full_sort_predict (beam search 64/64/64 + collect_scores -> [users, n_items] scores: what hopwise's evaluation loop
calls per batch) on bench_pgpr's graph, an evaluation batch of several ``beam_user_chunk`` chunks, with the chunks' searches one
after another (search_streams = 1) and dealt over three streams (the default)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench_pgpr

dev = torch.device("cuda", 0)
kg, ds, m = bench_pgpr.make_model(dev)
rng = np.random.default_rng(0)
n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if len(sys.argv) > 2:  # beam, e.g. 25,5,1 (hopwise's PGPR.yaml default; the bench line runs 64,64,64)
    m.beam_search_hop = [int(x) for x in sys.argv[2].split(",")]
if len(sys.argv) > 3:
    m.user_chunk = int(sys.argv[3])
users = torch.from_numpy(rng.integers(1, kg.n_users, n_users)).to(dev)
inter = {m.USER_ID: users}
out = {"users": n_users, "beam": list(m.beam_search_hop), "beam_user_chunk": m.user_chunk, "n_items": kg.n_items}
for streams in (1, 3, 1, 3):
    m.search_streams = streams
    m.full_sort_predict(inter)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s = m.full_sort_predict(inter)
    torch.cuda.synchronize()
    out.setdefault(f"ms_per_256_users_streams{streams}", []).append(round((time.perf_counter() - t0) / 3 / n_users * 256 * 1e3, 3))
print(json.dumps(out))
