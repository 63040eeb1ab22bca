#!/usr/bin/env python
"""One GEMM shape a few times (for rocprofv3 counter passes):  python tools/gemm_one.py M N K a_kc b_kc [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hopwise_amd import ops

M, N, K, akc, bkc = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def padded(r, c):
    return torch.randn((r, (c + 31) // 32 * 32), device=dev)[:, :c]


a = padded(M, K) if akc else padded(K, M)
b = padded(N, K) if bkc else padded(K, N)
out = padded(M, N)
for _ in range(reps):
    ops.gemm(a, b, bool(akc), bool(bkc), out=out)
torch.cuda.synchronize()
