#!/usr/bin/env python
"""ConvE (d=200 as a 10 x 20 image, 32 3x3 filters) and ConvKB on the Alibaba-iFashion shape, B = 2048: forward and
forward+backward of calculate_loss.  The 1-vs-all score product + BCE is shared with TuckER (ops.ScoreAllBceMean)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hopwise_amd.model as M
from tests.helpers import FakeDataset, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
nU, nI, nE, nR, B = 114738, 30041, 59157, 53, a.batch
rng = np.random.default_rng(0)
it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
    user_id=rng.integers(1, nU, B), item_id=rng.integers(1, nI, B), neg_item_id=rng.integers(1, nI, B),
    head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
    neg_tail_id=rng.integers(1, nE, B)).items()}
for name, d, extra in (("ConvE", 200, dict(embedding_shape=10, input_dropout=0.2, hidden_dropout=0.3, feature_dropout=0.2,
                                           label_smoothing=0.1, use_bias=True, hidden_size=10368)),
                       ("ConvKB", 64, {"out_channels": 32, "kernel_size": 1, "dropout_prob": 0.5, "lambda": 0.2})):
    torch.manual_seed(0)
    try:
        m = getattr(M, name)(make_config(dev, d, 1.0, **extra), FakeDataset(nU, nI, nE, nR)).to(dev)
    except Exception as e:  # config keys differ between the two models; report and go on
        print(name, "not constructed:", type(e).__name__, e)
        continue
    for mode in ("fwd", "fwd+bwd"):
        def one():
            if mode == "fwd":
                with torch.no_grad():
                    return m.calculate_loss(it)
            m.zero_grad(set_to_none=True)
            loss = m.calculate_loss(it)
            loss = sum(loss) if isinstance(loss, tuple) else loss
            loss.backward()
        for _ in range(2):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            one()
        torch.cuda.synchronize()
        print(f"{name:7s} d={d:3d} B={B} {mode:8s}: {(time.perf_counter() - t0) / a.reps * 1e3:8.2f} ms")
