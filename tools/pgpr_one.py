"""A few 256-user 64/64/64 beam searches on bench_pgpr.py's graph and model, nothing else: the program tools/pmc_pgpr.sh
profiles (one rocprofv3 --pmc pass per counter group)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_pgpr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
kg, ds, m = bench_pgpr.make_model(dev)
rng = np.random.default_rng(100)
for i in range(n):
    m.beam_search_device(torch.from_numpy(rng.integers(1, kg.n_users, 256)).to(dev))
torch.cuda.synchronize()
print("done")
