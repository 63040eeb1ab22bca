#!/usr/bin/env python
"""Full sort of the bilinear scorers (DistMult / ComplEx / Analogy: score = <query(h, r), tail>) as ONE product on the MFMA GEMM
(model.gemm_full_sort, the default) against the gather kernel (kge_full_sort_kernel / kge_full_sort_topk_kernel), Yelp2018-sized
candidate set: full_sort_predict and the whole evaluation step (scores -> history mask -> top-10)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hopwise_amd.model as M
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


nU, nI, nE, nR = 45920, 45539, 90962, 44
for name, d in (("ComplEx", 256), ("DistMult", 200), ("Analogy", 128)):
    torch.manual_seed(0)
    m = getattr(M, name)(make_config(dev, d, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    m.eval()
    rng = np.random.default_rng(0)
    for B in (256, 2048):
        users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
        hist = (torch.from_numpy(rng.integers(0, B, 50 * B)).to(dev), torch.from_numpy(rng.integers(1, nI, 50 * B)).to(dev))
        inter = {"user_id": users}
        out = {"model": name, "d": d, "B": B, "N": nI}
        with torch.no_grad():
            for tag, flag in (("gemm", True), ("kernel", False)):
                m.gemm_full_sort = flag
                out[f"full_sort_{tag}_ms"] = timed(lambda: m.full_sort_predict(inter))
                out[f"eval_top10_{tag}_ms"] = timed(lambda: m.full_sort_topk(inter, 10, hist))
            m.gemm_full_sort = True
            a = m.full_sort_predict(inter)
            from hopwise_amd import ops
            full = a.view(B, nI).clone()
            out["mask_ms"] = timed(lambda: ops.mask_scores_(full, hist))
            out["topk10_two_stage_ms"] = timed(lambda: ops.topk_rows(full, 10))
            out["topk10_one_stage_ms"] = timed(lambda: ops.topk_rows(full, 10, two_stage=False))
            m.gemm_full_sort = False
            b = m.full_sort_predict(inter).view_as(a)
            out["max_abs_diff_over_max_score"] = float((a - b).abs().max() / b.abs().max())
        print(json.dumps(out), flush=True)
