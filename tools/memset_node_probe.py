"""Is a memset NODE of a replayed HIP graph ordered with the kernel nodes around it?  (late round 3.)

The capture: [kernel: x <- 1] -> [hipMemsetAsync(x, 0)] -> [kernel: y <- x] -> [kernel: x <- 2], all on the capturing stream, with
x large enough for the fills to take a while.  Stream order says y == 0 after every replay.  y != 0 means the memset ran before
the first fill had finished (y sees 1) or after the copy had started; counts of both are printed.  Eager control: the same four
operations issued on a plain stream."""
import ctypes
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device("cuda", 0)


def ops(x, y, stream):
    x.fill_(1.0)
    rc = hip.hipMemsetAsync(x.data_ptr(), 0, x.numel() * 4, ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, rc
    y.copy_(x)
    x.fill_(2.0)


def run(n_floats, replays, graph):
    x = torch.empty(n_floats, device=dev)
    y = torch.empty(n_floats, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    bad = 0
    seen = set()
    if graph:
        with torch.cuda.stream(side):
            ops(x, y, side)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops(x, y, torch.cuda.current_stream())
    for _ in range(replays):
        if graph:
            g.replay()
        else:
            with torch.cuda.stream(side):
                ops(x, y, side)
            torch.cuda.current_stream().wait_stream(side)
        nz = y != 0
        if bool(nz.any()):
            bad += 1
            seen.update(y[nz].unique().tolist()[:4])
        # some eager work between the replays, as a training loop has
        _ = torch.empty(1 << 16, device=dev).normal_().abs().max()
    return bad, sorted(seen)


for n in (51200 // 4, 1 << 20, 1 << 24):
    for graph in (False, True):
        bad, seen = run(n, 300, graph)
        print(f"{'graph' if graph else 'eager'} x = {n * 4} bytes: y != 0 after {bad} of 300 runs, values seen {seen}")
sys.exit(0)
