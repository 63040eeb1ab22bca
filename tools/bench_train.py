#!/usr/bin/env python
"""Training-step throughput in the reference's own batch layout (shape (i)):
rec side 8 positives x 256 negatives tiled (2048 rows) + kg side 2048 triples x (1 pos + 1 neg) = 8192 scored triples
per calculate_loss; forward, forward+backward and forward+backward+dense Adam, against the CPU probe in BASELINE.md
(RotatE: 0.52 M triples/s forward, 0.24 M forward+backward on 8 Xeon cores)."""
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
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda", 0)
nU, nI, nE, nR = 6041, 2985, 13805, 13
rng = np.random.default_rng(0)
for name, d, K, big in (("TransE", 64, 1, 1), ("RotatE", 200, 256, 1), ("DistMult", 200, 1, 1), ("ComplEx", 256, 1, 1),
                        ("RotatE", 200, 1, 16)):
    cfg = make_config(dev, d, 1.0, train_neg_sample_args={"sample_num": K})
    torch.manual_seed(0)
    m = getattr(M, name)(cfg, FakeDataset(nU, nI, nE, nR)).to(dev)
    if os.environ.get("HW_OPTIM", "hw") == "hw":  # the engine's one-launch Adam with gradient sinks; "torch": torch.optim.Adam
        from hopwise_amd.optim import FusedAdam

        opt = FusedAdam(m.parameters(), lr=1e-3)
    else:
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    B = 2048 * big
    P = B // K
    users, pos = rng.integers(1, nU, P), rng.integers(1, nI, P)
    it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B),
        head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
        neg_tail_id=rng.integers(1, nE, B)).items()}
    triples = 4 * B
    out = []
    for mode in ("fwd", "fwd+bwd", "fwd+bwd+adam"):
        def one():
            if mode == "fwd":
                with torch.no_grad():
                    return m.calculate_loss(it)
            opt.zero_grad(set_to_none=True)
            l = m.calculate_loss(it)
            l.backward()
            if mode.endswith("adam"):
                opt.step()
            return l
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        out.append(f"{mode} {dt*1e3:7.3f} ms = {triples/dt/1e6:8.2f} M triples/s")
    print(f"{name:8s} d={d:3d} sample_num={K:3d} batch={B:5d}: " + " | ".join(out))
