"""The row-sharded path END TO END with the real HIP kernels and a real two-process exchange: two ranks share the box's
one GPU under the gloo backend (collectives staged through host memory, parallel._staged -- RCCL refuses two ranks on one
device; the RCCL calls themselves are exercised by tools/rccl_single_rank_check.py and timed by bench.py --gpus N).
Checks, against the plain unsharded model on the same data: the loss of every rank, the gradient of every shard (row
gradients travel to their owners), the all-reduced relation gradients, three FusedAdam steps, and the sharded full sort /
masked top-k (values and global ids exact)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
MODEL, DIM, NU, NI, NE, NR, B = "ComplEx", 64, 400, 300, 900, 7, 512


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank, step=0):
    g = np.random.default_rng(100 + rank + 17 * step)
    return {k: g.integers(1, hi, B) for k, hi in (
        ("user_id", NU), ("item_id", NI), ("neg_item_id", NI), ("head_id", NE), ("relation_id", NR - 1), ("tail_id", NE),
        ("neg_tail_id", NE))}


def _worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HW_CTYPES_ONLY"] = os.environ.get("HW_CTYPES_ONLY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import hopwise_amd.model as M
    from hopwise_amd import parallel as par
    from hopwise_amd.optim import FusedAdam
    from tests.helpers import FakeDataset, load_tables, make_config, random_tables

    tables = random_tables(MODEL, NU, NE, NR, DIM, seed=4)

    def build():
        return load_tables(getattr(M, MODEL)(make_config(dev, DIM, 1.0), FakeDataset(NU, NI, NE, NR)).to(dev), tables, dev)

    to_dev = lambda it: {k: torch.from_numpy(v).to(dev) for k, v in it.items()}
    try:
        # ---- loss + gradients: mean over ranks of the plain model's losses == what the sharded ranks compute -------------
        plain = build()
        losses = []
        for r in range(world):
            l = plain.calculate_loss(to_dev(_batch(r))) / world
            l.backward()
            losses.append(float(l.detach()) * world)
        sh = par.RowShardedKGE(build())
        loss = sh.calculate_loss(to_dev(_batch(rank)))
        loss.backward()
        sh.sync_grads()
        sh.check_overflow()
        assert abs(float(loss.detach()) - losses[rank]) <= 1e-6 * abs(losses[rank])
        for n in plain.USER_TABLES + plain.ENTITY_TABLES:
            want = getattr(plain, n).weight.grad[rank::world].cpu().numpy()
            got = getattr(sh, "shard_" + n).grad.cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6 * (np.abs(want).max() + 1e-12) + 1e-9, err_msg=n)
        for n in plain.RELATION_TABLES:
            want = getattr(plain, n).weight.grad.cpu().numpy()
            np.testing.assert_allclose(getattr(sh.model, n).weight.grad.cpu().numpy(), want, rtol=2e-4,
                                       atol=2e-6 * np.abs(want).max() + 1e-9, err_msg=n)
        # ---- three optimizer steps: FusedAdam on the shards (sinks, caller syncs) vs torch Adam on the full tables --------
        plain, sh = build(), par.RowShardedKGE(build())
        o1 = torch.optim.Adam(plain.parameters(), lr=1e-2)
        o2 = FusedAdam(sh.parameters(), lr=1e-2, grad_sinks="caller_syncs")
        for step in range(3):
            o1.zero_grad()
            sum(plain.calculate_loss(to_dev(_batch(r, step))) for r in range(world)).div(world).backward()
            o1.step()
            o2.zero_grad()
            sh.calculate_loss(to_dev(_batch(rank, step))).backward()
            sh.sync_grads()
            o2.step()
        sh.check_overflow()
        for n in plain.USER_TABLES + plain.ENTITY_TABLES:
            np.testing.assert_allclose(getattr(sh, "shard_" + n).detach().cpu().numpy(),
                                       getattr(plain, n).weight.detach()[rank::world].cpu().numpy(), rtol=1e-4, atol=3e-6, err_msg=n)
        # ---- sharded full sort == plain full sort --------------------------------------------------------------------------
        plain, sh = build(), par.RowShardedKGE(build())
        g = np.random.default_rng(9)  # same queries on every rank
        users = torch.from_numpy(g.integers(1, NU, 32)).to(dev)
        hist = (torch.arange(32, device=dev).repeat_interleave(5), torch.from_numpy(g.integers(1, NI, 160)).to(dev))
        kg = {"head_id": torch.from_numpy(g.integers(1, NE, 16)).to(dev), "relation_id": torch.from_numpy(g.integers(1, NR - 1, 16)).to(dev)}
        with torch.no_grad():
            assert torch.equal(sh.full_sort_predict({"user_id": users}), plain.full_sort_predict({"user_id": users}))
            assert torch.equal(sh.full_sort_predict_kg(kg), plain.full_sort_predict_kg(kg))
            for inter, h, is_kg in (({"user_id": users}, hist, False), (kg, None, True)):
                v0, i0 = plain.full_sort_topk(inter, 10, h, kg=is_kg)
                v1, i1 = sh.full_sort_topk(inter, 10, h, kg=is_kg)
                assert torch.equal(v0, v1) and torch.equal(i0, i1), is_kg
        sd = sh.full_state_dict()
        if rank == 0:
            for k, v in tables.items():
                assert np.array_equal(sd[k + ".weight"].cpu().numpy(), v), k
        torch.cuda.synchronize()
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_real_kernels(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
