#!/usr/bin/env python
"""cProfile of the host side of one KGE evaluation step (RotatE d = 200, 2 048 users x 2 985 items): the step is bound by its
launching thread there (tools/eval_host_enqueue.py); this says which Python it is (three descriptor builds, a dozen allocations,
eight stream look-ups per step)."""
import cProfile, pstats, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import hopwise_amd.model as M
from tests.helpers import FakeDataset, make_config
dev = torch.device("cuda", 0)
nU, nI, nE, nR = 6041, 2985, 13805, 13
torch.manual_seed(0)
m = M.RotatE(make_config(dev, 200, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev); m.eval()
rng = np.random.default_rng(0); B = 2048
users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
hist = (torch.from_numpy(rng.integers(0, B, 50 * B)).to(dev), torch.from_numpy(rng.integers(1, nI, 50 * B)).to(dev))
inter = {"user_id": users}
with torch.no_grad():
    for _ in range(5): m.full_sort_topk(inter, 10, hist)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): m.full_sort_topk(inter, 10, hist)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
