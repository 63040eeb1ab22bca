#!/usr/bin/env python
"""Timing of the (f)-row additions on one GPU: the widened KGE scorers (grouped forward at the headline shape and one
calculate_loss + backward step) and a PGPR REINFORCE episode (forward rollout and the stacked backward).

    python tools/bench_widened.py [--reps 20] > gpurun_out/widened.jsonl
One JSON line per measurement; algorithmic bytes follow DESIGN.md's convention (tail rows + ids + amortised query)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench_pgpr
from hopwise_amd import ops
from tests.helpers import FakeDataset, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--models", default="TransE,RotatE,TorusE,TransH,TransD,Analogy,HolE,RESCAL,TransR")
ap.add_argument("--no-pgpr", action="store_true")
ap.add_argument("--forward-only", action="store_true", help="grouped forward timings only (kernel tuning)")
a = ap.parse_args()
dev = torch.device("cuda", 0)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


import hopwise_amd.model as M

nU, nI, nE, nR = 6041, 2985, 13805, 13
B, K = 2048, 256
for name in a.models.split(","):
    d = 64 if name in ("RESCAL", "TransR") else 200
    torch.manual_seed(0)
    m = getattr(M, name)(make_config(dev, d, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    rng = np.random.default_rng(0)
    if name != "TransR":
        head, rel, tail = m._tables("rec")
        users = torch.randint(1, nU, (B,), device=dev)
        tails = torch.randint(1, nI, (B, 1 + K), device=dev)
        desc = ops.make_desc(name, head, rel, tail, margin=1.0, tail_rows=nI)
        ms = timed(lambda: ops.kge_score_grouped_raw(desc, users, None, nR - 1, tails), a.reps)
        nt = len(tail)
        abytes = (nt * d * 4 + 12) * B * (K + 1)
        print(json.dumps({"what": "grouped_forward", "model": name, "dim": d, "B": B, "K": K, "ms": ms,
                          "triples_per_s": B * (K + 1) / ms * 1e3, "alg_TB_per_s": abytes / ms / 1e9}), flush=True)
    if a.forward_only:
        continue
    nb = 4096
    inter = {k: torch.randint(1, hi, (nb,), device=dev) for k, hi in (
        ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1), ("tail_id", nE),
        ("neg_tail_id", nE))}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.calculate_loss(inter)
        loss.backward()
        opt.step()

    ms = timed(step, a.reps)
    print(json.dumps({"what": "train_step(loss+backward+adam)", "model": name, "dim": d, "rec": nb, "kg": nb, "ms": ms,
                      "triples_per_s": 4 * nb / ms * 1e3}), flush=True)

if a.forward_only or a.no_pgpr:
    sys.exit(0)
# PGPR REINFORCE episode
kg, ds, pm = bench_pgpr.make_model(dev)
pm.action_dropout = 0.5
for nu in (64, 512, 2048):
    users = torch.from_numpy(np.random.default_rng(1).integers(1, kg.n_users, nu)).to(dev)
    ep = pm.episode_device(users)
    ms_f = timed(lambda: pm.episode_device(users), max(3, a.reps // 4))
    ms_b = timed(lambda: pm._episode_backward(ep), max(3, a.reps // 4))
    print(json.dumps({"what": "pgpr_episode", "users": nu, "steps": 3, "ms_rollout": ms_f, "ms_backward": ms_b,
                      "users_per_s": nu / (ms_f + ms_b) * 1e3}), flush=True)
