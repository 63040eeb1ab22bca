#!/usr/bin/env python
"""The three-way bf16 split loop of gemm_f32_kernel (HW_GEMM_SPLIT=1, the default for K-contiguous x K-contiguous products)
next to the exact-FMA loop (HW_GEMM_SPLIT=0): error of both against a float64 product on edge shapes and on operands with a
wide dynamic range, then TFLOP/s on the TuckER / PGPR shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hopwise_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def padded(r, c, wide=False):
    ld = (c + 31) // 32 * 32
    t = torch.randn((r, ld), device=dev)
    if wide:
        t = t * torch.exp(3.0 * torch.randn((r, ld), device=dev))
    return t[:, :c]


def err(got, want, scale):
    # |error| relative to sum_k |a_ik b_kj|: the scale a floating-point dot product's error bound is stated on
    return float(((got.double() - want).abs() / scale.clamp_min(1e-300)).max())


def operands(M, N, K, akc, bkc, wide=False):
    a = padded(M, K, wide) if akc else padded(K, M, wide)
    b = padded(N, K, wide) if bkc else padded(K, N, wide)
    A = a.double() if akc else a.double().t()
    Bm = b.double().t() if bkc else b.double()
    return a, b, A, Bm


PERF_ONLY = "--perf-only" in sys.argv
worst = {"1": 0.0, "0": 0.0}
for wide in (() if PERF_ONLY else (False, True)):
    for (M, N, K) in ((4, 4, 4), (5, 7, 9), (128, 128, 32), (130, 131, 33), (64, 200, 200), (257, 513, 72), (300, 1000, 8),
                      (300, 1000, 40), (77, 256, 1234), (2048, 3000, 200), (4096, 256, 512), (2048, 200, 5000)):
        for akc in (True, False):
            for bkc in (True, False):
                a, b, A, Bm = operands(M, N, K, akc, bkc, wide)
                want, scale = A @ Bm, A.abs() @ Bm.abs()
                for mode in ("1", "0"):
                    os.environ["HW_GEMM_SPLIT"] = mode
                    e = err(ops.gemm(a, b, akc, bkc), want, scale)
                    worst[mode] = max(worst[mode], e)
                    assert e < 5e-6, (M, N, K, akc, bkc, wide, mode, e)
if not PERF_ONLY:
    print(f"max |err| / sum|a b|:  split {worst['1']:.3e}   exact-FMA loop {worst['0']:.3e}")
# epilogues: bias + ELU / ReLU, Philox dropout (same keep bits in both loops)
a, b, bias = padded(300, 400), padded(256, 400), torch.randn(256, device=dev)  # K-contiguous x K-contiguous
for kw in (() if PERF_ONLY else (dict(bias=bias, elu=True), dict(bias=bias, elu="relu"),
                                       dict(bias=bias, elu=True, drop_mode=2, seed=5, offset=3))):
    outs = []
    for mode in ("1", "0"):
        os.environ["HW_GEMM_SPLIT"] = mode
        outs.append(ops.gemm(a, b, True, True, **kw))
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-4, atol=1e-4)
    assert torch.equal(outs[0] == 0, outs[1] == 0) or kw.get("elu") == "relu"
if not PERF_ONLY:
    print("epilogues agree")

B, d = 2048, 200
shapes = []
for tag, N in (("rec", 144779), ("kg", 59157)):
    shapes += [(f"{tag} xE  [2048,200]x[{N},200]^T", B, N, d, True, True),
               (f"{tag} dX  [2048,{N}]x[{N},200]", B, d, N, True, False),
               (f"{tag} dE  [2048,{N}]^T x [2048,200]", N, d, B, False, False)]
shapes += [("pgpr l2 [1048576,512]x[256,512]^T", 1048576, 256, 512, True, True),
           ("pgpr actor [1048576,256]x[251,256]^T", 1048576, 251, 256, True, True),
           ("pgpr l2 [16384,512]x[256,512]^T", 16384, 256, 512, True, True),
           ("dW [16384,512]^T x [16384,400]", 512, 400, 16384, False, False),
           ("dXl [16384,512]x[512,400]", 16384, 400, 512, True, False),
           ("4096^3 kc/kc", 4096, 4096, 4096, True, True), ("4096^3 km/km", 4096, 4096, 4096, False, False)]
for name, M, Nn, K, akc, bkc in shapes:
    a, b, _, _ = operands(M, Nn, K, akc, bkc)
    out = padded(M, Nn)
    line = f"{name:42s}"
    for mode in (("1",) if PERF_ONLY else ("1", "0")):
        os.environ["HW_GEMM_SPLIT"] = mode
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
        line += f"  split={mode}: {ms:8.3f} ms {2.0 * M * Nn * K / ms / 1e9:6.1f} TF/s"
    print(line)
    del a, b, out
