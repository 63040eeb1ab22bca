#!/usr/bin/env python
"""The vocabulary log-probabilities of a CAFE run_program call (1 024 users x 13 nodes = 13 312 rows over 13 805 entities,
d = 100) piece by piece: the logits GEMM into a buffer with the natural row pitch (13 805 floats: rows start at odd 4-byte
offsets) against a padded pitch, the row log-softmax pass, and the GEMM that leaves per-stretch (max, sum exp) pairs + the
row finish instead of that pass."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from hopwise_amd import ops
from hopwise_amd import pgpr_ops as P

dev = torch.device("cuda", 0)
M, N, K = 13312, 13805, 100
torch.manual_seed(0)
x = torch.randn(M, K, device=dev) * 0.3
e = torch.randn(N, K, device=dev) * 0.3


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


nat = torch.empty((M, N), device=dev)
pad = ops.padded_rows(M, N, dev) if hasattr(ops, "padded_rows") else torch.empty((M, (N + 31) // 32 * 32), device=dev)[:, :N]
lp = torch.empty((M, (N + 31) // 32 * 32), device=dev)[:, :N]
print(f"gemm, natural pitch   {t_us(lambda: ops.gemm(x, e, out=nat)):8.1f} us")
print(f"gemm, padded pitch    {t_us(lambda: ops.gemm(x, e, out=pad)):8.1f} us")
print(f"log_softmax nat->pad  {t_us(lambda: P.log_softmax_rows(nat, out=lp)):8.1f} us")
print(f"log_softmax pad->pad  {t_us(lambda: P.log_softmax_rows(pad, out=lp)):8.1f} us")
print(f"fill_ (write only)    {t_us(lambda: pad.fill_(1.0)):8.1f} us   {M * N * 4 / 1e6:.0f} MB")
row_lse = torch.empty((M, 2), device=dev)
print(f"gemm + row lse        {t_us(lambda: ops.gemm_lse(x, e, out=pad, row_lse=row_lse)):8.1f} us   (product with the pairs in its epilogue + hw_lse_rows)")
ops.gemm(x, e, out=nat)
ref = torch.logsumexp(nat.double(), dim=1)
print("max |lse - float64 lse|:", ((row_lse[:, 0] + row_lse[:, 1]).double() - ref).abs().max().item())
