#!/usr/bin/env python
"""BASELINE.json configs[3]: ComplEx d = 256 on a Yelp2018-shaped KG with the user / entity tables row-sharded over the
ranks (owner = id % P, all-to-all lookup; hopwise_amd/parallel.py).  Stand-alone version of bench.py's `sharded` leg (which
is what the driver times); kept for experiments with the optimizer / sink knobs.  One training step = calculate_loss on this rank's
batch -> backward (row gradients travel back to their owners) -> all-reduce of the replicated relation tables -> Adam on
the local shards.

    python tools/bench_sharded.py                      # one rank
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sharded.py

Prints one JSON line on rank 0: aggregate triples/s, ms/step, and the bytes one rank exchanges per step."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from hopwise_amd import parallel
from hopwise_amd.data import synthetic
from hopwise_amd.model import ComplEx
from tests.helpers import FakeDataset, make_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192, help="rec positives and kg triples per rank per step")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
a = ap.parse_args()
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl")
dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
torch.cuda.set_device(dev)
nU, nI, nE, nR, _, _ = synthetic.SHAPES["yelp2018"]
D = 256
torch.manual_seed(0)  # same full tables on every rank before sharding
model = ComplEx(make_config(dev, D, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
sharded = parallel.RowShardedKGE(model).to(dev)
if os.environ.get("HW_OPTIM", "hw") == "hw":  # hopwise_amd.optim.FusedAdam: one launch, gradient clear folded in
    from hopwise_amd.optim import FusedAdam

    # the sharded lookup scatters the returned row gradients straight into the persistent .grad shards; sync_grads()
    # all-reduces the replicated relation tables' p.grad itself (what "caller_syncs" declares)
    opt = FusedAdam(sharded.parameters(), lr=1e-3, grad_sinks="caller_syncs" if int(os.environ.get("HW_SINKS", "1")) else False)
else:
    opt = torch.optim.Adam(sharded.parameters(), lr=1e-3, fused=bool(int(os.environ.get("HW_FUSED_ADAM", "1"))))
rng = np.random.default_rng(100 + rank)
B = a.batch


def batch():
    return {k: torch.from_numpy(rng.integers(1, hi, B)).to(dev) for k, hi in (
        ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1), ("tail_id", nE),
        ("neg_tail_id", nE))}


def step(b):
    opt.zero_grad(set_to_none=True)
    loss = sharded.calculate_loss(b)
    loss.backward()
    sharded.sync_grads()
    opt.step()
    return loss


batches = [batch() for _ in range(4)]
for i in range(a.warmup):
    step(batches[i % 4])
if world > 1:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    loss = step(batches[i % 4])
if world > 1:
    dist.barrier()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
# exchange volume of one rank and step: unique ids out, rows back (forward) and row gradients out (backward)
b0 = batches[0]
ent = torch.unique(torch.cat([b0[k] for k in ("item_id", "neg_item_id", "head_id", "tail_id", "neg_tail_id")])).numel()
usr = torch.unique(b0["user_id"]).numel()
remote = (world - 1) / world
xbytes = remote * ((ent + usr) * 8 + 2 * (ent + usr) * 2 * D * 4)
if rank == 0:
    print(json.dumps({
        "metric": "ComplEx d=256 training triples per second, user/entity tables row-sharded (Yelp2018 shape)",
        "value": world * 4 * B * a.steps / dt, "unit": "triples/s", "n_gpus": world, "steps": a.steps,
        "ms_per_step": dt / a.steps * 1e3, "batch_per_rank": {"rec": B, "kg": B}, "loss": float(loss.detach()),
        "unique_rows_per_step_per_rank": {"entity": ent, "user": usr},
        "exchange_bytes_per_step_per_rank": xbytes, "table_bytes_per_rank": 2 * (nU + nE) * D * 4 / world,
        "scaling": "weak", "parallelism": f"row-sharded x{world}"}))
if world > 1:
    dist.destroy_process_group()
