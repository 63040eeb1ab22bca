import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from hopwise_amd import ops
dev = torch.device("cuda", 0)
M, N = 13312, 13805
def t_us(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
pad = ops.padded_rows(M, N, dev)
for K in (64, 96, 100, 112, 128, 200, 256):
    x = torch.randn(M, K, device=dev) * 0.3
    e = torch.randn(N, K, device=dev) * 0.3
    a = t_us(lambda: ops.gemm(x, e, out=pad))
    b = t_us(lambda: ops.gemm_lse(x, e, out=pad))
    c = t_us(lambda: ops.gemm_rank(x, e, out=pad))
    print(f"K={K:4d}  gemm {a:7.1f} us   gemm_lse {b:7.1f} us   three-term {c:7.1f} us")
# M sweep at K=100
K = 100
e = torch.randn(N, K, device=dev) * 0.3
for M2 in (3328, 6656, 13312, 26624):
    x = torch.randn(M2, K, device=dev) * 0.3
    p2 = ops.padded_rows(M2, N, dev)
    print(f"M={M2:6d} gemm_lse {t_us(lambda: ops.gemm_lse(x, e, out=p2)):7.1f} us")
