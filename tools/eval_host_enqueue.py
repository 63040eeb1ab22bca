#!/usr/bin/env python
"""Host time of one evaluation step (full_sort_topk: query rows -> ranking product -> candidates -> exact re-rank) against its
device time, RotatE d = 200 and DistMult d = 200 at ml-1m-small's 2 985 items and Yelp2018's 45 539, 2 048 users per step: is the
step's wall time the device's or the launching thread's?"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import hopwise_amd.model as M
from tests.helpers import FakeDataset, make_config

dev = torch.device("cuda", 0)
for nU, nI, nE, nR in ((6041, 2985, 13805, 13), (45920, 45539, 90962, 44)):
    for name in ("RotatE", "DistMult"):
        torch.manual_seed(0)
        m = getattr(M, name)(make_config(dev, 200, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
        m.eval()
        rng = np.random.default_rng(0)
        B = 2048
        users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
        hist = (torch.from_numpy(rng.integers(0, B, 50 * B)).to(dev), torch.from_numpy(rng.integers(1, nI, 50 * B)).to(dev))
        inter = {"user_id": users}
        with torch.no_grad():
            for _ in range(3):
                m.full_sort_topk(inter, 10, hist)
            torch.cuda.synchronize()
            host = []
            for _ in range(20):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m.full_sort_topk(inter, 10, hist)
                host.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                m.full_sort_topk(inter, 10, hist)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 50
        print(json.dumps({"model": name, "items": nI, "host_ms_per_step": round(float(np.median(host)) * 1e3, 4),
                          "wall_ms_per_step_back_to_back": round(wall * 1e3, 4)}), flush=True)
