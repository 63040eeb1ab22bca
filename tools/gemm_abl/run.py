#!/usr/bin/env python
"""Time the ablation variants of the split GEMM loop (tools/gemm_abl/build.sh):  run.py M N K a_kc b_kc
Which part of a launch is matrix work, which is operand conversion / LDS traffic, which is the output store."""
import ctypes as C
import glob
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
M, N, K, akc, bkc = (int(v) for v in sys.argv[1:6])
BCE = len(sys.argv) > 6 and sys.argv[6] == "bce"  # the fused sigmoid + BCELoss epilogue (x . E^T only)
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def padded(r, c):
    return torch.randn((r, (c + 31) // 32 * 32), device=dev)[:, :c]


a = padded(M, K) if akc else padded(K, M)
b = padded(N, K) if bkc else padded(K, N)
out = padded(M, N)
if BCE:
    a.mul_(0.05)  # |scores| well below 16: the epilogue's unclamped form, as with trained embeddings
ws = torch.empty(64 << 20, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
order = ["base", "nostore", "nomma", "noput", "nofetch", "nosync", "noread", "mmaonly"]
libs = {os.path.basename(p)[8:-3]: p for p in glob.glob(os.path.join(HERE, "libgemm_*.so"))}
names = [n for n in order if n in libs] + sorted(set(libs) - set(order))
loaded = {n: C.CDLL(libs[n]) for n in names}


flag = torch.zeros(N, dtype=torch.uint8, device=dev)
loss_out = torch.zeros(1, device=dev)


def run(lib):
    if BCE:
        rc = lib.hw_gemm_f32_bce(P(a), C.c_int64(a.stride(0)), P(b), C.c_int64(b.stride(0)), None, P(out), C.c_int64(out.stride(0)),
                                 C.c_int64(M), C.c_int64(N), C.c_int64(K), P(flag), C.c_float(0.9), C.c_float(0.0005),
                                 P(loss_out), P(ws), C.c_int64(ws.numel()), None)
        assert rc == 0, rc
        return
    rc = lib.hw_gemm_f32(P(a), C.c_int64(a.stride(0)), akc, P(b), C.c_int64(b.stride(0)), bkc, P(out), C.c_int64(out.stride(0)),
                         None, C.c_int64(M), C.c_int64(N), C.c_int64(K), None, 0, None, 0, C.c_uint64(0), C.c_uint64(0),
                         P(ws), C.c_int64(ws.numel()), None)
    assert rc == 0, rc


# round-robin over the variants, several rounds: the card's clock drifts over the first seconds of load and a variant timed
# first reads slow (0.98 vs 0.83 ms for the same library in two positions)
times = {n: [] for n in names}
for rnd in range(6):
    for n in names:
        for _ in range(2):
            run(loaded[n])
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(10):
            run(loaded[n])
        eb.record()
        torch.cuda.synchronize()
        if rnd:
            times[n].append(ea.elapsed_time(eb) / 10)
A_ = a if akc else a.t()
B_ = b.t() if bkc else b
ref = A_.double() @ B_.double()
ref_loss = None
if BCE:
    ref_loss = (torch.nn.functional.softplus(ref) - 0.0005 * ref).mean().item()
    ref = (torch.sigmoid(ref) - 0.0005) / (M * N)
ref = ref.float()
scale = ref.abs().max().item()
for n in names:
    t = sorted(times[n])
    out.zero_()
    run(loaded[n])
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item() / scale
    extra = "" if ref_loss is None else f"   loss {loss_out.item():.9f} (float64 {ref_loss:.9f})"
    print(f"{n:10s} median {t[len(t) // 2]:7.3f} ms   min {t[0]:7.3f}   max |err| / max |c| = {err:.2e}{extra}", flush=True)
