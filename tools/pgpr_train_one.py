import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench_pgpr
dev = torch.device("cuda", 0)
kg, ds, m = bench_pgpr.make_model(dev)
m.train()
rng = np.random.default_rng(0)
users = torch.from_numpy(rng.integers(1, kg.n_users, int(os.environ.get("HW_TRAIN_B", "2048")))).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    loss = m.calculate_loss({m.USER_ID: users})
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("ms per step", (time.perf_counter()-t0)/20*1e3)
