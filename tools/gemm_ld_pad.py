#!/usr/bin/env python
"""Does the leading dimension of a GEMM operand / output matter?  Same products as TuckER's 1-vs-all step with the
[2048, N] score matrix stored with ld = N (odd: rows start at arbitrary 4-byte offsets, tile stores straddle cache lines,
and the K-contiguous reads of the backward fall back to dword loads) against ld rounded up to 32 floats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hopwise_amd import ops

dev = torch.device("cuda", 0)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B, d = 2048, 200
for N in (144779, 59157):
    E = torch.randn(N, d, device=dev)
    x = torch.randn(B, d, device=dev)
    for pad in (False, True):
        ld = (N + 31) // 32 * 32 if pad else N
        buf = torch.randn(B, ld, device=dev)
        S = buf[:, :N]
        gx, gE = torch.empty(B, d, device=dev), torch.empty(N, d, device=dev)
        t_f = timed(lambda: ops.gemm(x, E, True, True, out=S))          # scores = x E^T
        t_x = timed(lambda: ops.gemm(S, E, True, False, out=gx))        # dX = dS E
        t_e = timed(lambda: ops.gemm(S, x, False, False, out=gE))       # dE = dS^T x
        fl = 2.0 * B * N * d
        print(f"N={N} ld={ld}: scores {t_f:.3f} ms {fl/t_f/1e9:6.1f} TF/s | dX {t_x:.3f} ms {fl/t_x/1e9:6.1f} | "
              f"dE {t_e:.3f} ms {fl/t_e/1e9:6.1f}")
