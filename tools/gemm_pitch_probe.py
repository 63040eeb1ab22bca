#!/usr/bin/env python
"""Does the row pitch of a GEMM's OUTPUT matter beyond 16 bytes?  [13 312, 13 805] (CAFE logits) and [2 048, 45 539] (evaluation
scores) products into buffers whose rows are 16-, 64-, 128- and 256-byte aligned."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hopwise_amd import ops
dev = torch.device("cuda", 0)
def t_us(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, K in ((13312, 13805, 100), (2048, 45539, 400), (2048, 144779, 200)):
    x = torch.randn(M, K, device=dev) * 0.3
    e = torch.randn(N, K, device=dev) * 0.3
    for al in (4, 16, 32, 64):
        pitch = (N + al - 1) // al * al
        out = torch.empty((M, pitch), device=dev)[:, :N]
        a = t_us(lambda: ops.gemm(x, e, out=out))
        b = t_us(lambda: ops.gemm_lse(x, e, out=out))
        c = t_us(lambda: ops.gemm_rank(x, e, out=out))
        print(f"[{M}, {N}] K={K}  pitch {pitch} floats ({al * 4:3d}-byte rows)  gemm {a:7.1f}  gemm_lse {b:7.1f}  three-term {c:7.1f} us")
