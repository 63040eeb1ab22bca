#!/usr/bin/env python
"""Where the host time of one CAFE run_program_device call (256 users, idle device) goes, by stage of the call: wraps the
stage functions with a timer.  Round 5, final tree: 348 us per call -- collector 46, module struct table 44, interpreter call 41,
order draw 37, staging copy 29, plan kernel call 26, logits product call 25, module launch 22, excluded bits 12, glue ~60."""
import os, sys, time, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench_cafe
from hopwise_amd import ops
from hopwise_amd import pgpr_ops as P
from hopwise_amd.model import cafe as C
dev = torch.device("cuda", 0)
kg, ds, m, _ = bench_cafe.make_model(dev)
acc = collections.defaultdict(float)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t0
    setattr(obj, name, g)
wrap(m, "_order_tables"); wrap(m, "_module_structs"); wrap(m, "_excluded_bits"); wrap(m, "_order_tables_dev")
wrap(P, "cafe_plan"); wrap(P, "cafe_modules_forward_fused"); wrap(P, "cafe_execute"); wrap(P, "cafe_collect")
wrap(ops, "gemm_lse"); wrap(ops, "padded_rows"); wrap(C._Packed, "to", "packed.to")
users = torch.from_numpy(np.random.default_rng(1).integers(1, kg.n_users, 256))
for _ in range(5): m.run_program_device(users)
torch.cuda.synchronize(); acc.clear()
N = 50; tot = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.run_program_device(users); tot += time.perf_counter() - t0
print(f"total {tot / N * 1e6:.0f} us per call")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"  {k:28s} {v / N * 1e6:7.1f} us")
