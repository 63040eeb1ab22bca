#!/usr/bin/env python
"""Short products on few tiles: csrc/gemm_small.h against the 128-wide kernels (HW_GEMM_SMALL=0), HIP-event time per launch.
Shapes: CAFE's relation-module layers at 64 / 512 / 2048 paths (forward, dX, dW), PGPR's policy layers at 64 / 1024 users."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hopwise_amd import ops

dev = torch.device("cuda", 0)


def t_us(fn, reps=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows = []
for B in (64, 512, 2048):
    for name, M, N, K, akc, bkc in (("fc1 fwd", B, 256, 200, True, True), ("fc2 fwd", B, 200, 256, True, True),
                                    ("fc3 fwd", B, 100, 200, True, True), ("fc1 dX", B, 200, 256, True, False),
                                    ("fc1 dW", 256, 200, B, False, False), ("fc2 dW", 200, 256, B, False, False),
                                    ("policy l1", B, 512, 400, True, True), ("policy l2", B, 256, 512, True, True)):
        a = torch.randn((M, K) if akc else (K, M), device=dev)
        b = torch.randn((N, K) if bkc else (K, N), device=dev)
        out = torch.empty((M, N), device=dev)
        r = {"B": B, "product": name, "M": M, "N": N, "K": K}
        for small in ("0", "1"):
            os.environ["HW_GEMM_SMALL"] = small
            r["us_small" if small == "1" else "us_wide"] = round(t_us(lambda: ops.gemm(a, b, akc, bkc, out=out)), 2)
        rows.append(r)
        print(json.dumps(r), flush=True)
