#!/usr/bin/env python
"""CAFE run_program_device calls (1 024 users each, bench_cafe's model) issued one after another on one stream and dealt over
2 / 3 streams: ms per call.  The interpreter is latency-bound (short dependent launches), so calls on other streams can fill
the gaps -- if the host keeps up."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench_cafe

dev = torch.device("cuda", 0)
kg, ds, m, metapaths = bench_cafe.make_model(dev, 6)
n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
batches = [torch.from_numpy(rng.integers(1, kg.n_users, n_users)) for _ in range(12)]
out = {"users_per_call": n_users}
for streams in (1, 2, 3, 1, 2, 3):
    side = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    for st in side:
        with torch.cuda.stream(st):
            m.run_program_device(batches[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(3):
        for i, b in enumerate(batches):
            with torch.cuda.stream(side[i % streams]):
                r = m.run_program_device(b)
    torch.cuda.synchronize()
    out.setdefault(f"ms_per_call_streams{streams}", []).append(round((time.perf_counter() - t0) / 36 * 1e3, 3))
print(json.dumps(out))
