#!/usr/bin/env python
"""TuckER d=200 on the Alibaba-iFashion shape (BASELINE.json configs[4]): forward and forward+backward of
calculate_loss, achieved fp32 TFLOP/s against the 157.3 TFLOP/s matrix peak."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hopwise_amd.model import TuckER
from tests.helpers import FakeDataset, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
nU, nI, nE, nR, d, B = 114738, 30041, 59157, 53, 200, a.batch
cfg = make_config(dev, d, 1.0, label_smoothing=0.1, input_dropout=0.3, input_dropout1=0.4, input_dropout2=0.5)
torch.manual_seed(0)
m = TuckER(cfg, FakeDataset(nU, nI, nE, nR)).to(dev)
rng = np.random.default_rng(0)
it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
    user_id=rng.integers(1, nU, B), item_id=rng.integers(1, nI, B), head_id=rng.integers(1, nE, B),
    relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B)).items()}
flops_fwd = 0
for N in (nU + nI, nE):
    flops_fwd += 2 * B * d**3 + 2 * B * d * d + 2 * B * d * N
m.train()
for mode in ("fwd", "fwd+bwd"):
    for _ in range(2):
        m.zero_grad()
        l = m.calculate_loss(it)
        if mode != "fwd":
            l.backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        m.zero_grad(set_to_none=True)
        l = m.calculate_loss(it)
        if mode != "fwd":
            l.backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    fl = flops_fwd * (1 if mode == "fwd" else 3)
    print(f"TuckER d=200 iFashion B={B} {mode:8s}: {dt*1e3:8.2f} ms  {fl/dt/1e12:6.1f} TFLOP/s "
          f"({fl/dt/157.3e12*100:4.1f}% of fp32 MFMA peak)  {B*(nU+nI+nE)/dt/1e9:6.2f} G scores/s")
