#!/usr/bin/env python
"""The TuckER (iFashion shape) GEMMs as the engine issues them: operands with 16-byte aligned rows (the engine pads its
[B, N] buffers to a multiple of 32 floats), so the software-pipelined kernels run; HW_GEMM_NARROW=0 selects the 128-wide
tiles for the N = 200 products instead of the whole-row 16x16x4 kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hopwise_amd import ops

dev = torch.device("cuda", 0)


def padded(r, c):
    ld = (c + 31) // 32 * 32
    return torch.randn((r, ld), device=dev)[:, :c]


B, d = 2048, 200
for tag, N in (("rec", 144779), ("kg", 59157)):
    shapes = [
        (f"{tag} xE  [2048,200]x[{N},200]^T", padded(B, d), padded(N, d), True, True, B, N, d),
        (f"{tag} dX  [2048,{N}]x[{N},200]", padded(B, N), padded(N, d), True, False, B, d, N),
        (f"{tag} dE  [2048,{N}]^T x [2048,200]", padded(B, N), padded(B, d), False, False, N, d, B),
    ]
    for name, a, b, akc, bkc, M, Nn, K in shapes:
        for narrow in ("1", "0"):
            if Nn != d and narrow == "0":
                continue
            os.environ["HW_GEMM_NARROW"] = narrow
            out = padded(M, Nn)
            for _ in range(2):
                ops.gemm(a, b, akc, bkc, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                ops.gemm(a, b, akc, bkc, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            kind = "" if Nn != d else (" [narrow 16x16x4]" if narrow == "1" else " [128-wide tiles]")
            print(f"{name:42s}{kind:20s} {ms:8.3f} ms  {2.0 * M * Nn * K / ms / 1e9:7.1f} TFLOP/s")
