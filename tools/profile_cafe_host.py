#!/usr/bin/env python
"""cProfile of CAFE.run_program (256 users, bench_cafe's setup): where the host time goes."""
import cProfile
import os
import pstats
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench_cafe.py", "--reps", "1"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_cafe.py"))
m, users = g["m"], g["users"]
import torch

for _ in range(2):
    m.run_program(users)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.run_program(users)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
