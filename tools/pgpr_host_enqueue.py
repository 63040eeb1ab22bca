#!/usr/bin/env python
"""How long does the HOST take to enqueue one 256-user 64/64/64 search?  If that is close to the 3.7-4.2 ms the GPU takes per
search, the multi-stream figure is host-bound.  Times m.beam_search_device() calls (no synchronisation inside the timed loop)
and, next to it, the same number of searches with a synchronisation after each: host time per call against device time."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench_pgpr

dev = torch.device("cuda", 0)
kg, ds, m = bench_pgpr.make_model(dev)
rng = np.random.default_rng(0)
batches = [torch.from_numpy(rng.integers(1, kg.n_users, 256)).to(dev) for _ in range(8)]
for b in batches[:3]:
    m.beam_search_device(b)
torch.cuda.synchronize()
for overlap in (True, False):
    host = []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for b in batches:
        t0 = time.perf_counter()
        m.beam_search_device(b, overlap_actions=overlap)
        host.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print(f"overlap_actions={overlap}: host per call {np.mean(host[2:]) * 1e3:.2f} ms (first two {host[0] * 1e3:.2f}, {host[1] * 1e3:.2f}); "
          f"8 calls enqueued in {t_enq * 1e3:.1f} ms, all done after {t_tot * 1e3:.1f} ms")
