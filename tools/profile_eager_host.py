#!/usr/bin/env python
"""Where the HOST time of one eager training step goes (cProfile over 300 steps, RotatE d=200, reference batch layout)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hopwise_amd.model as M
from hopwise_amd.optim import FusedAdam
from tests.helpers import FakeDataset, make_config

dev = torch.device("cuda", 0)
nU, nI, nE, nR, d, K, B = 6041, 2985, 13805, 13, 200, 256, 2048
m = M.RotatE(make_config(dev, d, 1.0, train_neg_sample_args={"sample_num": K}), FakeDataset(nU, nI, nE, nR)).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
rng = np.random.default_rng(0)
P = B // K
users, pos = rng.integers(1, nU, P), rng.integers(1, nI, P)
it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
    user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B), head_id=rng.integers(1, nE, B),
    relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B), neg_tail_id=rng.integers(1, nE, B)).items()}


def step():
    opt.zero_grad(set_to_none=True)
    loss = m.calculate_loss(it)
    loss.backward()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
