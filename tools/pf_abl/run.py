#!/usr/bin/env python
"""Time the ablation variants of the fused PGPR policy kernel (tools/pf_abl/build.sh) on 1 M synthetic beam rows: which part of
the per-block time is matrix work, which is operand preparation, which is weight staging / barriers."""
import ctypes as C
import glob
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda", 0)
H1, H2, A, rows = 512, 256, 251, 1 << 20
nU, nE, nR = 6041, 13805, 13
g = torch.Generator(device="cpu").manual_seed(0)
tabs = [torch.randn(n, H1, generator=g).mul_(0.1).to(dev) for n in (nU, nU, nE, nU, nE, nR)]
zero = torch.zeros(H1, device=dev)
b1, b2 = torch.zeros(H1, device=dev), torch.zeros(H2, device=dev)
ba = torch.zeros(256, device=dev)
w2 = torch.randn(H2, H1, generator=g).mul_(0.05).to(dev)
wa = torch.randn(A, H2, generator=g).mul_(0.05).to(dev)
rng = np.random.default_rng(0)
rec = np.zeros((rows, 12), np.int32)
rec[:, 0] = np.repeat(rng.integers(1, nU, rows // 4096), 4096)          # uid: 4096 rows per user
rec[:, 1] = 3
rec[:, 2], rec[:, 3], rec[:, 4] = 12, 1, np.repeat(rng.integers(1, 2985, rows // 64), 64)   # hop 1: item
rec[:, 5], rec[:, 6], rec[:, 7] = rng.integers(1, 12, rows), 1, rng.integers(1, nE, rows)  # hop 2: entity
paths = torch.from_numpy(rec).to(dev)
n_rows = torch.tensor([rows], dtype=torch.int32, device=dev)
logits = torch.empty((rows, 256), device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
arr = (C.c_void_p * 6)(*[t.data_ptr() for t in tabs])
for lib_path in sorted(glob.glob(os.path.join(HERE, "libpf_*.so"))):
    lib = C.CDLL(lib_path)
    lib.hw_pgpr_policy_planes_bytes.restype = C.c_int64
    planes = torch.empty(int(lib.hw_pgpr_policy_planes_bytes(H1)), dtype=torch.uint8, device=dev)
    assert lib.hw_pgpr_policy_prepare(P(w2), H1, H2, P(wa), A, P(planes), None) == 0

    def run():
        rc = lib.hw_pgpr_policy_fused(arr, P(zero), P(b1), H1, P(planes), P(b2), P(ba), A, 1, P(paths), P(n_rows), C.c_int64(rows),
                                      3, None, None, C.c_uint64(1), C.c_uint64(2), C.c_uint64(3), P(logits), C.c_int64(256), None)
        assert rc == 0, rc

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for _ in range(10):
        run()
    eb.record()
    torch.cuda.synchronize()
    name = os.path.basename(lib_path)[6:-3]
    out = logits[:, :A].clone()
    if name == "base":
        ref = out
    note = ""
    if name.startswith("x_") and "ref" in globals():  # experiments that must not change a bit
        note = f"  bit-identical to base: {bool(torch.equal(out, ref))}  (max |diff| {(out - ref).abs().max().item():.3e})"
    print(f"{name:12s} {ea.elapsed_time(eb) / 10:8.3f} ms per launch of {rows} rows{note}", flush=True)
