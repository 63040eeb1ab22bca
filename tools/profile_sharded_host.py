#!/usr/bin/env python
"""cProfile of one row-sharded ComplEx training step at P = 1 (bench.py's `sharded` leg): where the host time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hopwise_amd import parallel
from hopwise_amd.model import ComplEx
from hopwise_amd.optim import FusedAdam
from tests.helpers import FakeDataset, make_config

dev = torch.device("cuda", 0)
nU, nI, nE, nR, Dm, B = 31668, 38048, 90961, 26, 256, 8192
torch.manual_seed(0)
model = ComplEx(make_config(dev, Dm, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
sharded = parallel.RowShardedKGE(model).to(dev)
opt = FusedAdam(sharded.parameters(), lr=1e-3, grad_sinks="caller_syncs")
rng = np.random.default_rng(0)
batch = {k: torch.from_numpy(rng.integers(1, hi, B)).to(dev) for k, hi in (
    ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1), ("tail_id", nE),
    ("neg_tail_id", nE))}


def step():
    opt.zero_grad(set_to_none=True)
    loss = sharded.calculate_loss(batch)
    loss.backward()
    sharded.sync_grads()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
