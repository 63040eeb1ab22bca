"""RotatE in the reference's batch layout (the bench's reference_layout leg), trained for --steps eager steps with a NaN
check of the loss, every parameter and every gradient after each step: says at which step and in which tensor a non-finite
value first appears."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hopwise_amd.model as M  # noqa: E402
from hopwise_amd.optim import FusedAdam  # noqa: E402
from tests.helpers import FakeDataset, make_config  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--poison", action="store_true", help="torch.empty returns NaN / max-int filled memory")
args = ap.parse_args()
if args.poison:
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
dev = torch.device("cuda", 0)
nU, nI, nE, nR, K, B, DIM = 6041, 2985, 13805, 13, 256, 2048, 200
rng = np.random.default_rng(0)
torch.manual_seed(0)
m = M.RotatE(make_config(dev, DIM, 1.0, train_neg_sample_args={"sample_num": K}), FakeDataset(nU, nI, nE, nR)).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
pos_n = B // K
users, pos = rng.integers(1, nU, pos_n), rng.integers(1, nI, pos_n)
it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
    user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B),
    head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
    neg_tail_id=rng.integers(1, nE, B)).items()}
names = dict(m.named_parameters())
if args.graph:
    from hopwise_amd.driver import GraphedTrainStep

    g = GraphedTrainStep(m, opt, it)
for i in range(args.steps):
    if args.graph:
        loss = g(it)
    else:
        opt.zero_grad(set_to_none=True)
        loss = m.calculate_loss(it)
        loss.backward()
        bad = [n for n, p in names.items() if p.grad is not None and not torch.isfinite(p.grad).all()]
        if bad:
            print("step", i, "non-finite grad in", bad, "loss", float(loss))
            for n in bad:
                gq = names[n].grad
                idx = (~torch.isfinite(gq)).nonzero()
                print(n, "rows", idx[:, 0].unique()[:20].tolist(), "count", idx.shape[0])
            sys.exit(1)
        opt.step()
    if not torch.isfinite(loss).all():
        print("step", i, "loss", float(loss))
        sys.exit(1)
    bad = [n for n, p in names.items() if not torch.isfinite(p).all()]
    if bad:
        print("step", i, "non-finite parameter", bad)
        for n in bad:
            idx = (~torch.isfinite(names[n])).nonzero()
            print(n, tuple(names[n].shape), "rows", idx[:, 0].unique()[:20].tolist(), "cols", idx[:, 1].unique()[:40].tolist(),
                  "count", idx.shape[0])
        sys.exit(1)
print("ok", args.steps, "steps, loss", float(loss))
