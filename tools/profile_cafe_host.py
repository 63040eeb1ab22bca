#!/usr/bin/env python
"""cProfile of CAFE.run_program_device (256 users, bench_cafe's setup): where the host time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench_cafe

dev = torch.device("cuda", 0)
kg, ds, m, metapaths = bench_cafe.make_model(dev)
users = torch.from_numpy(np.random.default_rng(1).integers(1, kg.n_users, int(os.environ.get("HW_CAFE_USERS", "1024"))))
for _ in range(2):
    m.run_program_device(users)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.run_program_device(users)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
