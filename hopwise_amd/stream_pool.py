"""Side streams that really run beside each other.

A HIP stream is served by one of a handful of hardware queues (four by default on this runtime, ``GPU_MAX_HW_QUEUES``), handed
out as streams are first used.  Two streams on one queue take turns: the packets of both sit in one ring, and the barrier a
stream's own ordering needs holds back the other's.  Which of torch's pooled streams share a queue depends on what the process
created before -- measured on MI355X (tools/stream_probe.py): of eight fresh streams, three pairs and one triple shared, and a
three-stream PGPR evaluation ran 5.2 or 4.8 ms per 256 users depending on which three it had been given.

So the streams are probed once: two single-thread spin kernels (``torch.cuda._sleep``) on two streams take as long as one when
the streams sit on different queues and twice that when they share one.  :func:`concurrent_streams` returns streams that are
pairwise concurrent with each other and with the stream(s) the caller names (its current stream), from a per-device set of
candidates; probes are cached, a call costs nothing after the first.  At most ``GPU_MAX_HW_QUEUES`` streams can be mutually
concurrent: beyond that the least conflicting candidates fill the list."""

from __future__ import annotations

import time

import torch

_CANDIDATES = 8
_SPIN_CYCLES = 400_000  # ~0.2 ms: long against launch jitter, short enough for a few dozen probes


class _DevicePool:
    def __init__(self, device):
        self.device = device
        self.cands = [torch.cuda.Stream(device=device) for _ in range(_CANDIDATES)]
        self._shared = {}
        self._single = None

    def _run(self, a, b):
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(_SPIN_CYCLES)
        with torch.cuda.stream(b):
            torch.cuda._sleep(_SPIN_CYCLES)
        torch.cuda.synchronize(self.device)
        return time.perf_counter() - t0

    def shares_queue(self, a, b):
        if a.cuda_stream == b.cuda_stream:
            return True
        key = (min(a.cuda_stream, b.cuda_stream), max(a.cuda_stream, b.cuda_stream))
        hit = self._shared.get(key)
        if hit is None:
            if self._single is None:
                self._run(self.cands[0], self.cands[0])  # first use of the stream, clocks up
                self._single = min(self._run(self.cands[0], self.cands[0]) for _ in range(2)) / 2
            self._run(a, b)
            t = min(self._run(a, b) for _ in range(2))
            hit = self._shared[key] = t > 1.5 * self._single
        return hit

    def pick(self, n, beside):
        chosen = []
        # 1. candidates on queues of their own; 2. candidates that share a queue with the caller's stream(s) only -- the caller
        # usually waits while its side streams work; 3. more streams wanted than queues: the remaining candidates, in order
        for avoid_caller in (True, False):
            for c in self.cands:
                if len(chosen) == n:
                    return chosen
                if c in chosen:
                    continue
                if not any(self.shares_queue(c, x) for x in chosen + (beside if avoid_caller else [])):
                    chosen.append(c)
        for c in self.cands:
            if len(chosen) == n:
                break
            if c not in chosen:
                chosen.append(c)
        while len(chosen) < n:
            chosen.append(torch.cuda.Stream(device=self.device))
        return chosen


_POOLS = {}
_PICKS = {}


def concurrent_streams(n, device=None, beside=None):
    """``n`` streams on ``device`` that run concurrently with each other and with the streams in ``beside`` (default: the
    caller's current stream).  The same arguments return the same streams.  Inside a stream capture, or without a GPU, plain
    new streams are returned (no probing there)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda" or n <= 0:
        return []
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if beside is None:
        beside = [torch.cuda.current_stream(device)]
    if torch.cuda.is_current_stream_capturing():
        return [torch.cuda.Stream(device=device) for _ in range(n)]
    key = (device.index, n, tuple(s.cuda_stream for s in beside))
    hit = _PICKS.get(key)
    if hit is None:
        if not hasattr(torch.cuda, "_sleep"):  # the probe's spin kernel is a private torch helper: without it, plain streams
            hit = _PICKS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
            return hit
        pool = _POOLS.get(device.index)
        if pool is None:
            pool = _POOLS[device.index] = _DevicePool(device)
        hit = _PICKS[key] = pool.pick(n, list(beside))
    return hit
