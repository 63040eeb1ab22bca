#!/usr/bin/env python
"""The two-waves-per-SIMD policy kernel experiment (tools/pf_abl/policy8.hip, variants built by build8.sh) against the product
kernel compiled into the same library, on 1 M synthetic beam rows: logits compared bit for bit, time per launch alternating
product / experiment."""
import ctypes as C
import glob
import json
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
b1, b2 = torch.randn(H1, generator=g).mul_(0.1).to(dev), torch.randn(H2, generator=g).mul_(0.1).to(dev)
ba = torch.zeros(256, device=dev)
ba[:A] = torch.randn(A, generator=g).mul_(0.1).to(dev)
w2 = torch.randn(H2, H1, generator=g).mul_(0.05).to(dev)
wa = torch.randn(A, H2, generator=g).mul_(0.05).to(dev)
rng = np.random.default_rng(0)
rec = np.zeros((rows, 12), np.int32)
rec[:, 0] = np.repeat(rng.integers(1, nU, rows // 4096), 4096)          # uid: 4096 rows per user
rec[:, 1] = 3
rec[:, 2], rec[:, 3], rec[:, 4] = 12, 1, np.repeat(rng.integers(1, 2985, rows // 64), 64)   # hop 1: item
rec[:, 5], rec[:, 6], rec[:, 7] = rng.integers(-1, 12, rows), 1, rng.integers(1, nE, rows)  # hop 2: entity (some self loops)
paths = torch.from_numpy(rec).to(dev)
n_rows = torch.tensor([rows], dtype=torch.int32, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
arr = (C.c_void_p * 6)(*[t.data_ptr() for t in tabs])


def load(path):
    lib = C.CDLL(path)
    lib.hw_pgpr_policy_planes_bytes.restype = C.c_int64
    lib.pf8_f32_bytes.restype = C.c_int64
    planes = torch.empty(int(lib.hw_pgpr_policy_planes_bytes(H1)), dtype=torch.uint8, device=dev)
    f32 = torch.empty(int(lib.pf8_f32_bytes(H1)), dtype=torch.uint8, device=dev)
    assert lib.hw_pgpr_policy_prepare(P(w2), H1, H2, P(wa), A, P(planes), None) == 0
    assert lib.pf8_prepare_f32(P(w2), H1, P(wa), A, P(f32), None) == 0

    def product(out):
        rc = lib.hw_pgpr_policy_fused(arr, P(zero), P(b1), H1, P(planes), P(b2), P(ba), A, 1, P(paths), P(n_rows), C.c_int64(rows), 3,
                                      None, None, C.c_uint64(1), C.c_uint64(2), C.c_uint64(3), P(out), C.c_int64(256), None)
        assert rc == 0, rc

    def experiment(out):
        rc = lib.pf8_policy_fused(arr, P(zero), P(b1), H1, P(planes), P(f32), P(b2), P(ba), A, P(paths), P(n_rows), C.c_int64(rows),
                                  C.c_uint64(1), C.c_uint64(2), C.c_uint64(3), P(out), C.c_int64(256), None)
        assert rc == 0, rc

    return lib, product, experiment, (planes, f32)


def timed(fn, out, reps=10):
    for _ in range(3):
        fn(out)
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for _ in range(reps):
        fn(out)
    eb.record()
    torch.cuda.synchronize()
    return round(ea.elapsed_time(eb) / reps, 4)


if __name__ == "__main__":
    la, lb = torch.empty((rows, 256), device=dev), torch.empty((rows, 256), device=dev)
    for lib_path in sorted(glob.glob(os.path.join(HERE, "libpf8_*.so"))):
        name = os.path.basename(lib_path)[7:-3]
        lib, product, experiment, keep = load(lib_path)
        la.fill_(7), lb.fill_(7)
        product(la), experiment(lb)
        torch.cuda.synchronize()
        out = {"variant": name, "bit_identical": bool(torch.equal(la, lb)), "product_ms": [], "experiment_ms": []}
        for _ in range(2):
            out["product_ms"].append(timed(product, la))
            out["experiment_ms"].append(timed(experiment, lb))
        print(json.dumps(out), flush=True)
