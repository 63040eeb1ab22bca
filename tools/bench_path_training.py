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
    # the same step replayed from a HIP graph (driver.GraphedTrainStep: the Philox offsets of the policy dropout, the action
    # dropout and the action draws live on the device during the capture), FusedAdam as in driver.graphed_trainer
    from hopwise_amd.driver import GraphedTrainStep
    from hopwise_amd.optim import FusedAdam

    fopt = FusedAdam(m.parameters(), lr=1e-4)
    it = {m.USER_ID: users}

    def fstep():
        fopt.zero_grad(set_to_none=True)
        loss = m.calculate_loss(it)
        loss.backward()
        fopt.step()

    ms_f = timed(fstep, 20)
    g = GraphedTrainStep(m, fopt, it)
    ms_g = timed(lambda: g(it), 50)
    g.check_flags()
    out[f"pgpr_B{B}"].update(ms_per_step_fused_adam=round(ms_f, 3), ms_per_step_graph_replay=round(ms_g, 3),
                             users_per_s_graph_replay=round(B / ms_g * 1e3))
    del g, fopt
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
    # the same device half replayed from its capture (driver.GraphedStagedTrainStep; the host half here hands over the same
    # batch: what is timed is the replay path -- host tensors into the static buffers, one graph launch, counters)
    from hopwise_amd.driver import make_graphed_step
    from hopwise_amd.optim import FusedAdam

    m._get_batch_by_user = lambda users, pos=pos, neg=neg_t.cpu().numpy(): (0, pos, neg)
    fopt = FusedAdam(m.parameters(), lr=1e-4)
    it = {m.USER_ID: torch.from_numpy(pos[:, 0].copy())}  # user ids as the loader hands them over: on the host

    def fstep():
        fopt.zero_grad()
        reg, rank = m.calculate_loss(it)
        (reg + rank).backward()
        fopt.step()

    ms_f = timed(fstep, 20)
    g = make_graphed_step(m, fopt, None)
    ms_g = timed(lambda: g(it), 50)
    out[f"cafe_B{B}"].update(ms_per_step_fused_adam=round(ms_f, 3), ms_per_step_graph_replay=round(ms_g, 3),
                             paths_per_s_graph_replay=round(B / ms_g * 1e3), captures=g.n_captured)
    del g, fopt
print(json.dumps(out), flush=True)

# TuckER (configs[4] width) and ConvE: eager trainer step against its graph replay (three / four dropouts per side, Philox offsets
# on the device during the capture)
del m
torch.cuda.empty_cache()
import hopwise_amd.model as M
from hopwise_amd.driver import GraphedTrainStep
from hopwise_amd.optim import FusedAdam
from tests.helpers import FakeDataset, make_config

nU, nI, nE, nR, Bt = 6041, 2985, 13805, 13, 2048
for name, cfg in (("TuckER", dict(embedding_size=200, label_smoothing=0.1, input_dropout=0.3, input_dropout1=0.4, input_dropout2=0.5)),):
    torch.manual_seed(0)
    c = make_config(dev, cfg.pop("embedding_size"), 1.0, **cfg)
    tm = getattr(M, name)(c, FakeDataset(nU, nI, nE, nR)).to(dev).train()
    it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        user_id=rng.integers(1, nU, Bt), item_id=rng.integers(1, nI, Bt), head_id=rng.integers(1, nE, Bt),
        relation_id=rng.integers(1, nR - 1, Bt), tail_id=rng.integers(1, nE, Bt)).items()}
    topt = FusedAdam(tm.parameters(), lr=1e-4)

    def tstep():
        topt.zero_grad(set_to_none=True)
        loss = tm.calculate_loss(it)
        loss.backward()
        topt.step()

    ms_e = timed(tstep, 20)
    g = GraphedTrainStep(tm, topt, it)
    ms_g = timed(lambda: g(it), 50)
    out[f"{name.lower()}_B{Bt}"] = {"ms_per_step_eager": round(ms_e, 3), "ms_per_step_graph_replay": round(ms_g, 3)}
    del g, topt, tm
    torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
