#!/usr/bin/env python
"""Which HIP streams run beside each other?  Two single-thread spin kernels (torch.cuda._sleep) on two streams take T when the
streams sit on different hardware queues and 2 T when they share one.  Prints the pair matrix for 8 streams + the default one,
then times the 3-stream PGPR evaluation on a concurrent triple and on a colliding one."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch

dev = torch.device("cuda", 0)
torch.cuda.init()
CY = 2_000_000  # ~1 ms
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(8)]
for s in streams:
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
torch.cuda.synchronize()


def pair_ms(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(CY)
    with torch.cuda.stream(b):
        torch.cuda._sleep(CY)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


one = pair_ms(streams[1], streams[1]) / 2
print("one sleep: %.2f ms" % one)
n = len(streams)
M = np.zeros((n, n))
for i in range(n):
    for j in range(i + 1, n):
        M[i, j] = M[j, i] = pair_ms(streams[i], streams[j]) / one
np.set_printoptions(precision=1, suppress=True, linewidth=200)
print("pair time / single time (1 = concurrent, 2 = serialized); row 0 = default stream")
print(M)
