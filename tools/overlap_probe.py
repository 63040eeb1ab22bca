#!/usr/bin/env python
"""Does alternating the configs[1] batches over two HIP streams (each with its own id / score buffers and loss chain) raise the
batch rate?  One generation of 2048 workgroups per launch leaves a ramp and a tail that a second stream could fill."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hopwise_amd import ops
from hopwise_amd.data import synthetic

dev = torch.device("cuda", 0)
kg = synthetic.make("ml-1m-small", seed=2024)
DIM, B, K = 200, 2048, 256
g = torch.Generator(device="cpu").manual_seed(2024)
xav = lambda rows: (torch.randn(rows, DIM, generator=g) * (2.0 / (rows + DIM)) ** 0.5).to(dev)
user_re, user_im, ent_re, ent_im = xav(kg.n_users), xav(kg.n_users), xav(kg.n_entities), xav(kg.n_entities)
theta = (torch.rand(kg.n_relations, DIM, generator=g) * 6.28).to(dev)
desc = ops.make_desc("RotatE", (user_re, user_im), (theta,), (ent_re, ent_im), margin=1.0, tail_rows=kg.n_items)
rng = np.random.default_rng(1)
pick = rng.integers(0, len(kg.inter_users), B)
users = torch.from_numpy(kg.inter_users[pick]).to(dev)
pos = torch.from_numpy(kg.inter_items[pick]).to(dev)
uptr, uidx = synthetic.used_csr(kg.inter_users, kg.inter_items, kg.n_users)
uptr, uidx = torch.from_numpy(uptr).to(dev), torch.from_numpy(uidx).to(dev)
N = 2000
for S in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(S)]
    chains = [ops.SelfAdvChain(B, dev) for _ in range(S)]
    tids, outs, losses = [], [], []
    for _ in range(S):
        t = torch.empty((B, 1 + K), dtype=torch.int64, device=dev)
        t[:, 0] = pos
        tids.append(t)
        outs.append(torch.empty((B, 1 + K), dtype=torch.float32, device=dev))
        losses.append(torch.empty(1, dtype=torch.float32, device=dev))
    torch.cuda.synchronize()

    def run(n, base):
        for i in range(n):
            s = i % S
            with torch.cuda.stream(streams[s]):
                chains[s].launch(desc, users, None, kg.ui_relation, tids[s], uptr, uidx, users, kg.n_items, 7, base + i, 1.0,
                                 out=outs[s], loss=losses[s])
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                chains[s].flush()

    run(200, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(N, 1000)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams={S}: {dt / N * 1e6:6.1f} us per batch ({B * (1 + K) * N / dt / 1e9:5.2f} G triples/s), host issue {t_host / N * 1e6:5.1f} us per batch",
          flush=True)
