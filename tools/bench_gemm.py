#!/usr/bin/env python
"""TFLOP/s of the fp32 MFMA GEMM on the shapes the path uses (PGPR policy layers, TuckER products)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hopwise_amd import ops

dev = torch.device("cuda", 0)
shapes = [
    ("pgpr l1   [131072,400]x[512,400]^T", 131072, 400, 512, True, True),
    ("pgpr l2   [131072,512]x[256,512]^T", 131072, 512, 256, True, True),
    ("pgpr act  [131072,256]x[251,256]^T", 131072, 256, 251, True, True),
    ("tucker rW [2048,200]x[200,40000]", 2048, 200, 40000, True, False),
    ("tucker xE [2048,200]x[144779,200]^T", 2048, 200, 144779, True, True),
    ("tucker xE kg [2048,200]x[59157,200]^T", 2048, 200, 59157, True, True),
    ("tucker dX [2048,144779]x[144779,200]", 2048, 144779, 200, True, False),
    ("tucker dE [2048,144779]^T x [2048,200]", 144779, 2048, 200, False, False),
    ("tucker dW [2048,200]^T x [2048,40000]", 200, 2048, 40000, False, False),
    ("tucker dr [2048,40000]x[200,40000]^T", 2048, 40000, 200, True, True),
    ("square 4096", 4096, 4096, 4096, True, True),
]
def padded(r, c):
    """[r, c] view of a buffer whose rows are padded to 32 floats -- what the engine does with its own [B, N] buffers
    (N = 144 779 is odd: unpadded rows start at arbitrary 4-byte offsets and the GEMM falls back to its unpipelined loop)."""
    return torch.randn((r, (c + 31) // 32 * 32), device=dev)[:, :c]


for name, M, K, N, akc, bkc in shapes:
    a = padded(*((M, K) if akc else (K, M)))
    b = padded(*((N, K) if bkc else (K, N)))
    out = padded(M, N)
    for _ in range(2):
        ops.gemm(a, b, akc, bkc, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, akc, bkc, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:40s} {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
