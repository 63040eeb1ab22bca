#!/usr/bin/env python
"""CAFE inference on the ml-1m-small-shaped CKG (north_star: PGPR / CAFE traversal): run_program (the neural program
executor over the metapath layout tree, cafe.py:406-447,790-857) for a batch of users, split into its device part
(_run_programs: relation modules on the MFMA GEMM, vocabulary log-softmax, program interpreter kernel) and the host
part (result collection in the reference's order; the reference runs the whole thing per user in Python).

    python tools/bench_cafe.py [--users 256] [--reps 3] [--metapaths 6]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench_pgpr
from hopwise_amd.data import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--metapaths", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda", 0)

from hopwise_amd.model import CAFE  # noqa: E402

kg = synthetic.make("ml-1m-small", seed=2024, n_triples=178_000)
ds = bench_pgpr._DS(kg)
names = {"[PAD]": 0, ds.ui_relation: kg.ui_relation}
for r in range(1, kg.n_relations):
    if r != kg.ui_relation:
        names[f"rel{r}"] = r
ds.field2token_id = {"relation_id": names}
# user -[UI]-> item -rel-> entity -rel-> item: the synthetic KG stores every triple in both directions under one id
rels = [n for n in names if n.startswith("rel")][: a.metapaths]
metapaths = [[[None, "user"], [ds.ui_relation, "entity"], [r, "entity"], [r, "entity"]] for r in rels]
cfg = dict(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", ENTITY_ID_FIELD="entity_id",
           RELATION_ID_FIELD="relation_id", HEAD_ENTITY_ID_FIELD="head_id", TAIL_ENTITY_ID_FIELD="tail_id",
           device=dev, path_constraint=metapaths, rank_weight=10.0, deep_module=True, use_dropout=True,
           topk_candidates=10, sample_size=500, topk_paths=25, max_user_trials=50, load_embeddings=True)
torch.manual_seed(2024)
m = CAFE(cfg, ds).to(dev)
m.eval()
users = torch.from_numpy(np.random.default_rng(1).integers(1, kg.n_users, a.users))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps, out


t_all, (scores, paths) = timed(lambda: m.run_program(users))
orders = [m._bfs_program() for _ in range(a.users)]
programs = [[(par, rel) for par, rel, _ in o] for o in orders]
ks = [[5] * len(o) for o in orders]
t_dev, res = timed(lambda: m._run_programs(users, programs, ks, grow_excluded=True, exclude=True))
n_paths = sum(int(c) for per in res for c, _, _ in per)
print(json.dumps({"what": "cafe_run_program", "users": a.users, "metapaths": len(metapaths),
                  "nodes_per_user": len(programs[0]), "paths_per_call": n_paths, "ms_total": t_all * 1e3,
                  "ms_run_programs": t_dev * 1e3, "users_per_s": a.users / t_all, "paths_per_s": n_paths / t_all,
                  "recommended_items": int(torch.isfinite(scores).sum())}))
