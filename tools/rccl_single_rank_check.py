#!/usr/bin/env python
"""Put the RCCL collectives of the row-sharded path on the stream of a ONE-GPU box: a world-size-1 "nccl" process group
with HW_FORCE_COLLECTIVES=1, so all_to_all_single (int32 counts / ids, fp32 rows, row gradients), all_reduce and
all_gather really go through RCCL (self send / recv).  Checks loss, gradients and the sharded full-sort top-k against
the plain model.  The multi-rank exchange itself is covered by tests/test_dist_gloo.py and timed by bench.py --gpus N."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HW_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
import torch
import torch.distributed as dist

from hopwise_amd import parallel as par
from hopwise_amd.model import ComplEx
from tests.helpers import FakeDataset, load_tables, make_config, random_tables

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
dim, nU, nI, nE, nR, B = 64, 400, 300, 900, 7, 512
tables = random_tables("ComplEx", nU, nE, nR, dim, seed=4)


def build():
    return load_tables(ComplEx(make_config(dev, dim, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev), tables, dev)


rng = np.random.default_rng(6)
inter = {k: torch.from_numpy(rng.integers(1, hi, B)).to(dev) for k, hi in (
    ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1), ("tail_id", nE),
    ("neg_tail_id", nE))}
plain, sh = build(), par.RowShardedKGE(build())
lp = plain.calculate_loss(inter)
lp.backward()
ls = sh.calculate_loss(inter)
ls.backward()
sh.sync_grads()
sh.check_overflow()
np.testing.assert_allclose(ls.item(), lp.item(), rtol=1e-6)
for n in plain.USER_TABLES + plain.ENTITY_TABLES:
    a, b = getattr(sh, "shard_" + n).grad.cpu().numpy(), getattr(plain, n).weight.grad.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6 * (np.abs(b).max() + 1e-12) + 1e-9, err_msg=n)
users = torch.from_numpy(rng.integers(1, nU, 32)).to(dev)
hist = (torch.arange(32, device=dev).repeat_interleave(5), torch.from_numpy(rng.integers(1, nI, 160)).to(dev))
v0, i0 = plain.full_sort_topk({"user_id": users}, 10, hist)
v1, i1 = sh.full_sort_topk({"user_id": users}, 10, hist)
assert torch.equal(v0, v1) and torch.equal(i0, i1)
assert torch.equal(sh.full_sort_predict({"user_id": users}), plain.full_sort_predict({"user_id": users}))
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl single-rank check ok: all_to_all_single / all_reduce / all_gather ran through RCCL; loss", float(ls))
