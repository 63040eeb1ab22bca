#!/usr/bin/env python
"""One training step (zero_grad -> calculate_loss -> backward -> Adam) replayed from a HIP graph, in the reference's
batch layout (SURVEY.md 8(d) config 2 shape (i): 8 users x 256 negatives tiled + 2048 kg triples x (1 pos + 1 neg)) or
at --rows positives per side with one negative each.  Run under rocprofv3 --kernel-trace --stats for the per-kernel
split of the replayed step.

    python tools/bench_train_graphed.py [--model RotatE] [--dim 200] [--rows 0] [--reps 200]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hopwise_amd.model as M
from hopwise_amd.driver import GraphedTrainStep
from tests.helpers import FakeDataset, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="RotatE")
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--rows", type=int, default=0, help="0: reference layout (2048 rec rows = 8 x 256, 2048 kg rows)")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--optim", default="fused", choices=["fused", "torch"], help="hopwise_amd.optim.FusedAdam / torch Adam")
a = ap.parse_args()
dev = torch.device("cuda", 0)
nU, nI, nE, nR = 6041, 2985, 13805, 13
rng = np.random.default_rng(0)
K = 256 if a.rows == 0 else 1
B = 2048 if a.rows == 0 else a.rows
cfg = make_config(dev, a.dim, 1.0, train_neg_sample_args={"sample_num": K})
torch.manual_seed(0)
m = getattr(M, a.model)(cfg, FakeDataset(nU, nI, nE, nR)).to(dev)
if a.optim == "fused":
    from hopwise_amd.optim import FusedAdam

    opt = FusedAdam(m.parameters(), lr=1e-3)
else:
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
P = B // K
users, pos = rng.integers(1, nU, P), rng.integers(1, nI, P)
it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
    user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B),
    head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
    neg_tail_id=rng.integers(1, nE, B)).items()}
step = GraphedTrainStep(m, opt, it)
for _ in range(10):
    step(it)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    step(it)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print(json.dumps({"what": "graphed_train_step", "model": a.model, "dim": a.dim, "optim": a.optim, "rec_rows": B, "kg_rows": B,
                  "ms": ms, "triples_per_s": 4 * B / ms * 1e3, "loss": float(step.loss)}))
