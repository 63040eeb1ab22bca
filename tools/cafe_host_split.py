import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench_cafe
from hopwise_amd.model import cafe as C
from hopwise_amd import pgpr_ops as P
dev = torch.device("cuda", 0)
kg, ds, m, metapaths = bench_cafe.make_model(dev)
users = torch.from_numpy(np.random.default_rng(1).integers(1, kg.n_users, 256))
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
wrap(m, "_bfs_order", "bfs_order")
wrap(m, "_run_programs", "run_programs(total)")
wrap(m, "_module_structs", "module_structs")
wrap(m, "_excluded_bits", "excluded_bits")
wrap(P, "cafe_modules_forward_fused", "modules_fused_call")
wrap(P, "cafe_execute", "cafe_execute_call")
wrap(P, "cafe_collect", "cafe_collect_call")
wrap(P, "log_softmax_rows", "log_softmax_call")
wrap(C._Packed, "to", "packed.to")
for _ in range(3): m.run_program_device(users)
torch.cuda.synchronize(); T.clear()
N = 50
t0 = time.perf_counter()
for _ in range(N): m.run_program_device(users)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("per call: host %.3f ms, wall %.3f ms" % (host / N * 1e3, wall / N * 1e3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print("  %-24s %.3f ms" % (k, v / N * 1e3))
print("distinct orders cached:", len(m._layout()["orders"]))
