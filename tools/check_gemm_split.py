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


def err(got, want, a, b):
    # |error| relative to sum_k |a_ik b_jk|: the scale a floating-point dot product's error bound is stated on
    scale = (a.double().abs() @ b.double().abs().t()).clamp_min(1e-300)
    return float(((got.double() - want).abs() / scale).max())


worst = {"1": 0.0, "0": 0.0}
for wide in (False, True):
    for (M, N, K) in ((1, 4, 4), (5, 7, 9), (128, 128, 32), (130, 131, 33), (64, 200, 200), (257, 513, 72), (300, 1000, 8),
                      (300, 1000, 40), (77, 256, 1234), (2048, 3000, 200), (4096, 256, 512)):
        a, b = padded(M, K, wide), padded(N, K, wide)
        want = a.double() @ b.double().t()
        for mode in ("1", "0"):
            os.environ["HW_GEMM_SPLIT"] = mode
            e = err(ops.gemm(a, b, True, True), want, a, b)
            worst[mode] = max(worst[mode], e)
            assert e < 5e-6, (M, N, K, wide, mode, e)
print(f"max |err| / sum|a b|:  split {worst['1']:.3e}   exact-FMA loop {worst['0']:.3e}")
# epilogues: bias + ELU / ReLU, Philox dropout (same keep bits in both loops)
a, b, bias = padded(300, 400), padded(256, 400), torch.randn(256, device=dev)
for kw in (dict(bias=bias, elu=True), dict(bias=bias, elu="relu"), dict(bias=bias, elu=True, drop_mode=2, seed=5, offset=3)):
    outs = []
    for mode in ("1", "0"):
        os.environ["HW_GEMM_SPLIT"] = mode
        outs.append(ops.gemm(a, b, True, True, **kw))
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-4, atol=1e-4)
    assert torch.equal(outs[0] == 0, outs[1] == 0) or kw.get("elu") == "relu"
print("epilogues agree")

B, d = 2048, 200
shapes = [(f"{tag} xE  [2048,200]x[{N},200]^T", B, N, d) for tag, N in (("rec", 144779), ("kg", 59157))]
shapes += [("pgpr l2 [1048576,512]x[256,512]^T", 1048576, 256, 512), ("pgpr actor [1048576,256]x[251,256]^T", 1048576, 251, 256),
           ("pgpr l2 [16384,512]x[256,512]^T", 16384, 256, 512), ("4096^3", 4096, 4096, 4096)]
for name, M, Nn, K in shapes:
    a, b, out = padded(M, K), padded(Nn, K), padded(M, Nn)
    line = f"{name:42s}"
    for mode in ("1", "0"):
        os.environ["HW_GEMM_SPLIT"] = mode
        for _ in range(2):
            ops.gemm(a, b, True, True, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            ops.gemm(a, b, True, True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        line += f"  split={mode}: {ms:8.3f} ms {2.0 * M * Nn * K / ms / 1e9:6.1f} TF/s"
    print(line)
    del a, b, out
