#!/usr/bin/env python
"""End to end without the reference installed: a KG (synthetic ml-1m-small shape, or hopwise atomic files) -> device
samplers -> RSKG batches -> a KGE model trained with the HIP-graph step -> full-sort evaluation with the fused masked
top-k -> Recall@10 / NDCG@10 on held-out interactions (leave-20 %-out per user, metrics on the host).

    python tools/example_train_eval.py [--model RotatE] [--epochs 3] [--atomic DIR NAME]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hopwise_amd.model as M
from hopwise_amd.data import synthetic
from hopwise_amd.driver import KGBatcher, train_epoch_graphed
from hopwise_amd.optim import FusedAdam
from hopwise_amd.sampler import KGSampler, Sampler
from tests.helpers import FakeDataset, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="RotatE")
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--atomic", nargs=2, metavar=("DIR", "NAME"))
a = ap.parse_args()
dev = torch.device("cuda", 0)

if a.atomic:
    from hopwise_amd.data.kg_files import KGFiles

    ds = KGFiles(a.atomic[0], a.atomic[1], kg_reverse_r=True)
    nU, nI, nE, nR = ds.user_num, ds.item_num, ds.entity_num, ds.relation_num
    iu, ii, heads, rels, tails = ds.inter_feat["user_id"], ds.inter_feat["item_id"], ds.head_entities, ds.relations, ds.tail_entities
else:
    kg = synthetic.make("ml-1m-small", seed=1, n_inter=200_000)
    nU, nI, nE, nR = kg.n_users, kg.n_items, kg.n_entities, kg.n_relations
    iu, ii, heads, rels, tails = kg.inter_users, kg.inter_items, kg.heads, kg.relations, kg.tails

rng = np.random.default_rng(0)
test = rng.random(len(iu)) < 0.2
import types

train_ds = types.SimpleNamespace(uid_field="user_id", iid_field="item_id", user_num=nU, item_num=nI, entity_num=nE,
                                 inter_feat={"user_id": iu[~test], "item_id": ii[~test]}, head_entities=heads,
                                 tail_entities=tails)
rec = Sampler("train", train_ds, device=dev).set_phase("train")
kgs = KGSampler(train_ds, device=dev)
torch.manual_seed(0)
model = getattr(M, a.model)(make_config(dev, a.dim, 2.0), FakeDataset(nU, nI, nE, nR)).to(dev)
opt = FusedAdam(model.parameters(), lr=5e-3)  # torch.optim.Adam(..., capturable=True) works too, ~2x the step time
batches = KGBatcher(iu[~test], ii[~test], heads, rels, tails, rec, kgs, train_batch_size=a.batch, device=dev)
step = None
for ep in range(a.epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, step = train_epoch_graphed(model, batches, opt, step=step)
    rec.check_keys = kgs.check_keys = False  # ids validated by the first epoch: drop the per-call device sync
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"epoch {ep}: loss {loss / len(batches):.4f}  {len(batches)} steps in {dt:.2f} s "
          f"({2 * a.batch * len(batches) / dt / 1e6:.1f} M triples/s incl. sampling and batching)")

# evaluation: top-10 over all items with the train interactions masked, 2048 users per call
model.eval()
truth = {}
for u, i in zip(iu[test], ii[test]):
    truth.setdefault(int(u), set()).add(int(i))
users = np.array(sorted(truth))
order = np.argsort(iu[~test], kind="stable")
tu, ti = iu[~test][order], ii[~test][order]
ptr = np.searchsorted(tu, np.arange(nU + 1))
hits = ndcg = 0.0
t0 = time.perf_counter()
with torch.no_grad():
    for s in range(0, len(users), 2048):
        chunk = users[s : s + 2048]
        rows = np.concatenate([np.full(ptr[u + 1] - ptr[u], r) for r, u in enumerate(chunk)])
        cols = np.concatenate([ti[ptr[u] : ptr[u + 1]] for u in chunk])
        hist = (torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev))
        _, idx = model.full_sort_topk({"user_id": torch.from_numpy(chunk).to(dev)}, 10, hist)
        idx = idx.cpu().numpy()
        for r, u in enumerate(chunk):
            rel = np.array([int(x) in truth[int(u)] for x in idx[r]], float)
            hits += rel.sum() / min(len(truth[int(u)]), 10)
            dcg = (rel / np.log2(np.arange(2, 12))).sum()
            ndcg += dcg / (1.0 / np.log2(np.arange(2, 2 + min(len(truth[int(u)]), 10)))).sum()
torch.cuda.synchronize()
print(f"eval: {len(users)} users x {nI} items in {time.perf_counter() - t0:.2f} s   Recall@10 {hits / len(users):.4f}   "
      f"NDCG@10 {ndcg / len(users):.4f}")
