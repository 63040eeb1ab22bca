#!/usr/bin/env python
"""time(K) of the fp32 MFMA GEMM at a fixed output shape: the intercept is the per-tile prologue + epilogue cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hopwise_amd import ops

dev = torch.device("cuda", 0)
for M, N in ((131072, 512), (2048, 40000)):
    out = torch.empty((M, N), device=dev)
    for K in (16, 64, 128, 256, 400, 800, 1600, 3200):
        a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        for _ in range(2):
            ops.gemm(a, b, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm(a, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"M={M} N={N} K={K:5d} {ms:8.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
