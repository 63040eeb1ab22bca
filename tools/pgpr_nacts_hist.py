import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench_pgpr
dev=torch.device("cuda",0)
kg, ds, m = bench_pgpr.make_model(dev)
rng=np.random.default_rng(100)
users=torch.from_numpy(rng.integers(1, kg.n_users, 256)).to(dev)
res=m.beam_search_device(users, keep_buffers=True)
n=int(res.n_rows_per_hop[2].item())
na=res.buffers["n_acts"][:n].cpu().numpy()
h=np.bincount(np.minimum((na+31)//32, 8), minlength=9)
print(json.dumps({"rows": n, "mean_n_acts": float(na.mean()), "tiles_needed_hist(ceil(n_acts/32))": h.tolist(), "mean_tiles": float(((na+31)//32).mean()), "frac_full_251": float((na==251).mean())}))
# if sorted into tiles of 32 rows: tiles needed = ceil(max/32) per group
s=np.sort(na); g=s[: (n//32)*32].reshape(-1,32).max(1); print("sorted mean tiles per 32-row group", float(((g+31)//32).mean()), "unsorted", float((((na[: (n//32)*32].reshape(-1,32).max(1))+31)//32).mean()))
