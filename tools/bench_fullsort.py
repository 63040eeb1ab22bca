#!/usr/bin/env python
"""Evaluation path: full sort -> history mask -> top-k, unfused ([B, N] matrix written and re-read)
against the fused kernel that never materialises it.  RotatE d = 200, ml-1m-small and Yelp2018-sized candidate sets."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hopwise_amd.model as M
from hopwise_amd import ops
from tests.helpers import FakeDataset, make_config

dev = torch.device("cuda", 0)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, (nU, nI, nE, nR) in (("ml-1m-small", (6041, 2985, 13805, 13)), ("yelp2018", (45920, 45539, 90962, 44))):
    torch.manual_seed(0)
    m = M.RotatE(make_config(dev, 200, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    m.eval()
    rng = np.random.default_rng(0)
    for B in (256, 2048):
        users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
        hist = (torch.from_numpy(rng.integers(0, B, 50 * B)).to(dev), torch.from_numpy(rng.integers(1, nI, 50 * B)).to(dev))
        inter = {"user_id": users}
        with torch.no_grad():
            def unfused():
                s = m.full_sort_predict(inter).view(B, nI)
                ops.mask_scores_(s, hist)
                return ops.topk_rows(s, 10)

            t_full = timed(lambda: m.full_sort_predict(inter))
            t_unf = timed(unfused)
            t_fus = timed(lambda: m.full_sort_topk(inter, 10, hist))
        tail_bytes = nI * 2 * 200 * 4
        print(json.dumps({"dataset": name, "B": B, "N": nI, "full_sort_ms": t_full, "unfused_eval_ms": t_unf,
                          "fused_eval_ms": t_fus, "scores_per_s_fused": B * nI / t_fus * 1e3,
                          "full_sort_alg_GBps": (tail_bytes * ((B + 3) // 4) / 1.0 + B * nI * 4) / t_full / 1e6,
                          "matrix_bytes_avoided": B * nI * 4}), flush=True)
