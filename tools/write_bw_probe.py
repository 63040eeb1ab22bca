#!/usr/bin/env python
"""How fast can this GPU WRITE?  A [2048, 45568] fp32 matrix (373 MB: the evaluation's score matrix) filled by torch, copied
device-to-device, and written by hw_fill_bytes -- the ceiling the ranking product's epilogue is measured against."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda", 0)
x = torch.empty((2048, 45568), device=dev)
y = torch.randn((2048, 45568), device=dev)
def t_us(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
nb = x.numel() * 4
for name, fn, bytes_ in (("fill_", lambda: x.fill_(1.0), nb), ("copy_ (read + write)", lambda: x.copy_(y), 2 * nb), ("zero_", lambda: x.zero_(), nb)):
    us = t_us(fn)
    print(f"{name:24s} {us:8.1f} us  {bytes_ / us / 1e6:6.2f} TB/s")
