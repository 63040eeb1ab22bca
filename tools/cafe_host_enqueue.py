#!/usr/bin/env python
"""Host time of one CAFE run_program_device call (no synchronisation inside the timed loop) against its device time."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench_cafe

dev = torch.device("cuda", 0)
kg, ds, m, _ = bench_cafe.make_model(dev)
for n_users in (256, 1024, 2048):
    users = torch.from_numpy(np.random.default_rng(1).integers(1, kg.n_users, n_users))
    for _ in range(3):
        m.run_program_device(users)
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(10):
        torch.cuda.synchronize()  # an idle device: the call's own host time, not a wait for a queue slot or a staging buffer
        t0 = time.perf_counter()
        m.run_program_device(users)
        host.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print(f"{n_users} users: host per call {np.mean(host[2:]) * 1e3:.2f} ms (min {min(host) * 1e3:.2f}); 10 calls enqueued in "
          f"{t_enq * 1e3:.1f} ms, all done after {t_tot * 1e3:.1f} ms")
