#!/usr/bin/env python
"""One CAFE symbolic-network training step (forward + backward + Adam) repeated, 512 paths per batch: the program
tools/sum_train_trace.py profiles (pass 'cafe' as its marker)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench_cafe
dev = torch.device("cuda", 0)
kg, ds, m, metapaths = bench_cafe.make_model(dev, 6)
m.train()
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mp = m.metapaths[0]
pos = np.stack([rng.integers(1, kg.n_users, B), rng.integers(1, kg.n_items, B), rng.integers(1, kg.n_entities, B),
                rng.integers(1, kg.n_items, B)], 1).astype(np.int64)
pos_t, neg_t = torch.from_numpy(pos).to(dev), torch.from_numpy(rng.integers(1, kg.n_items, B)).to(dev)
opt = torch.optim.Adam(m.model.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    reg, rank = m.model(mp, pos_t, neg_t, m.embeddings)
    (reg + rank * m.rank_weight).backward()
    opt.step()
if len(sys.argv) > 2 and sys.argv[2] == "graph":  # the same step through driver.GraphedStagedTrainStep (host ids stay on the host)
    from hopwise_amd.driver import make_graphed_step
    from hopwise_amd.optim import FusedAdam
    neg = neg_t.cpu().numpy()
    m._get_batch_by_user = lambda users: (0, pos, neg)
    g = make_graphed_step(m, FusedAdam(m.parameters(), lr=1e-4), None)
    it = {m.USER_ID: torch.from_numpy(pos[:, 0].copy())}
    step = lambda: g(it)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 20 * 1e3)
