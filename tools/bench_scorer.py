#!/usr/bin/env python
"""Micro-benchmark of the grouped scorer kernel alone (interleaved variants in one process, rule 24).

    python tools/bench_scorer.py [--model RotatE] [--dim 200] [--reps 30] VARIANT[,VARIANT...]
VARIANT = name:ENV=VAL:ENV=VAL...   (environment seen by the launcher, e.g. "xcd:HW_GROUPED_VARIANT=1")
Reports ms and algorithmic TB/s for (a) the ml-1m-small tables with negatives over items, (b) negatives over
all 13 805 entities, (c) a 2.6 M-row (4.2 GB) entity table.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hopwise_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="?", default="base")
ap.add_argument("--model", default="RotatE")
ap.add_argument("--dim", type=int, default=200)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, K, d = 2048, 256, a.dim
nU, nI, nE, nR, nBig = 6041, 2985, 13805, 13, 2_621_440
two = a.model in ("RotatE", "ComplEx")
g = torch.Generator(device="cpu").manual_seed(0)
mk = lambda r: (torch.randn(r, d, generator=g) * 0.05).to(dev)
head = (mk(nU), mk(nU)) if two else (mk(nU),)
rel = (mk(nR), mk(nR)) if a.model == "ComplEx" else (mk(nR),)
ent = (mk(nE), mk(nE)) if two else (mk(nE),)
big = tuple(torch.randn(nBig, d, device=dev) * 0.05 for _ in range(2 if two else 1))
users = torch.randint(1, nU, (B,), device=dev)
cases = {
    "rows256": (ent, torch.randint(1, 256, (B, 1 + K), device=dev)),
    "items": (ent, torch.randint(1, nI, (B, 1 + K), device=dev)),
    "entities": (ent, torch.randint(1, nE, (B, 1 + K), device=dev)),
    "yelp91k": (tuple(torch.randn(90962, d, device=dev) * 0.05 for _ in range(2 if two else 1)),
                torch.randint(1, 90962, (B, 1 + K), device=dev)),
    "hbm4GB": (big, torch.randint(1, nBig, (B, 1 + K), device=dev)),
}
nt = 2 if two else 1
abytes = (nt * d * 4 + 12 + ((len(head) + len(rel)) * d * 4 + 16) / (K + 1)) * B * (K + 1)
variants = []
for v in a.variants.split(","):
    parts = v.split(":")
    variants.append((parts[0], dict(p.split("=") for p in parts[1:])))
res = {}
for rnd in range(a.rounds):
    for name, env in variants:
        for k in list(os.environ):
            if k.startswith("HW_"):
                del os.environ[k]
        os.environ.update(env)
        for cname, (tail, tidx) in cases.items():
            desc = ops.make_desc(a.model, head, rel, tail, margin=1.0)
            for _ in range(3):
                ops.kge_score_grouped_raw(desc, users, None, nR - 1, tidx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops.kge_score_grouped_raw(desc, users, None, nR - 1, tidx)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, cname), []).append(e0.elapsed_time(e1) / a.reps)
ref = {}
for (name, cname), v in res.items():
    ms = float(np.median(v))
    print(f"{name:14s} {cname:9s} median {ms*1e3:8.1f} us  min {min(v)*1e3:8.1f} us   {abytes/ms/1e9:7.2f} TB/s algorithmic")
# correctness cross-check between variants
outs = {}
for name, env in variants:
    for k in list(os.environ):
        if k.startswith("HW_"):
            del os.environ[k]
    os.environ.update(env)
    tail, tidx = cases["entities"]
    outs[name] = ops.kge_score_grouped_raw(ops.make_desc(a.model, head, rel, tail, margin=1.0), users, None, nR - 1, tidx).cpu().numpy()
base = outs[variants[0][0]]
for name, o in outs.items():
    print(name, "max |diff| vs first variant:", float(np.abs(o - base).max()))
