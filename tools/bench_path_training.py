#!/usr/bin/env python
"""Training steps of the path-reasoning models, on bench_pgpr / bench_cafe's ml-1m-shaped graph and models:
  * PGPR: calculate_loss (one REINFORCE episode of max_path_len = 3 steps per user: action dropout, policy network, Categorical
    draw, rewards, update()'s loss) + backward (hand-written, csrc/pgpr_train.hip) + Adam step;
  * CAFE: calculate_loss (the host-side path sampler of the reference + the symbolic network on the device) + backward + Adam.
ms per step and users per second, with the host-side share of the CAFE step (its sampler is host code in the reference too)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench_cafe
import bench_pgpr

dev = torch.device("cuda", 0)


def timed(step, reps):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {}
kg, ds, m = bench_pgpr.make_model(dev)
m.train()
rng = np.random.default_rng(0)
for B in (64, 1024, 4096):
    users = torch.from_numpy(rng.integers(1, kg.n_users, B)).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.calculate_loss({m.USER_ID: users})
        loss.backward()
        opt.step()

    ms = timed(step, 20)
    out[f"pgpr_B{B}"] = {"ms_per_step": round(ms, 3), "users_per_s": round(B / ms * 1e3), "episode_steps": m.max_path_len}
print(json.dumps(out), flush=True)
del m
torch.cuda.empty_cache()
kg, ds, m, metapaths = bench_cafe.make_model(dev, 6)
m.train()
m.rng = np.random.default_rng(3)
for B in (64, 512, 4096):
    # the device side of calculate_loss: SymbolicNetwork.forward on one (metapath, positive paths, negative items) batch.  The
    # paths are id-valid random walks: the reference's host-side sampler (cafe.py:143-198, host code here as well) needs
    # the user's top candidates to be reachable, which random embeddings on a synthetic graph do not give
    mp = m.metapaths[0]
    pos = np.stack([rng.integers(1, kg.n_users, B), rng.integers(1, kg.n_items, B), rng.integers(1, kg.n_entities, B),
                    rng.integers(1, kg.n_items, B)], 1).astype(np.int64)
    pos_t, neg_t = torch.from_numpy(pos).to(dev), torch.from_numpy(rng.integers(1, kg.n_items, B)).to(dev)
    opt = torch.optim.Adam(m.model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        reg, rank = m.model(mp, pos_t, neg_t, m.embeddings)
        (reg + rank * m.rank_weight).backward()
        opt.step()

    ms = timed(step, 20)
    out[f"cafe_B{B}"] = {"ms_per_step": round(ms, 3), "paths_per_s": round(B / ms * 1e3), "metapath_levels": len(mp) - 1}
print(json.dumps(out), flush=True)
