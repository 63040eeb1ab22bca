#!/usr/bin/env python
"""dX = dS . E and dE = dS^T . x of one TuckER side, back to back on one stream vs concurrently on two (product library)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from hopwise_amd import ops

dev = torch.device("cuda", 0)
B, d = 2048, 200
N = int(sys.argv[1]) if len(sys.argv) > 1 else 144779


def padded(r, c):
    return torch.randn((r, (c + 31) // 32 * 32), device=dev)[:, :c]


dS, E, X = padded(B, N), padded(N, d), padded(B, d)
dX, dE = padded(B, d), padded(N, d)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    ops.gemm(dS, E, True, False, out=dX)
    ops.gemm(dS, X, False, False, out=dE)


def concurrent():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ops.gemm(dS, E, True, False, out=dX)
    with torch.cuda.stream(s2):
        ops.gemm(dS, X, False, False, out=dE)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


for name, fn in (("serial", serial), ("concurrent", concurrent), ("serial", serial), ("concurrent", concurrent)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for _ in range(10):
        fn()
    eb.record()
    torch.cuda.synchronize()
    print(f"{name:12s} {ea.elapsed_time(eb) / 10:7.3f} ms", flush=True)
