"""Device-side routing of the row-sharded lookup (csrc/shard.hip) against oracle/shard_plan.py, bit for bit, and a
multi-owner exchange simulated on ONE GPU: P virtual ranks, the all-to-alls done by hand with tensor slicing, so that the
P > 1 code path of the kernels (owner-major bitmap, per-peer counts, capacity padding) runs on hardware even though the
box has one card.  The real multi-process exchange is tests/test_dist_gloo.py (CPU stand-ins) and bench.py --gpus N (RCCL).
"""

import numpy as np
import pytest
import torch

from oracle import shard_plan as osp

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("T", [1, 2])
@pytest.mark.parametrize("P,n_rows,shape", [(1, 97, (40,)), (2, 53, (7, 5)), (3, 1000, (4096,)), (8, 90962, (5, 8192)),
                                            (8, 45920, (8192,)), (4, 33, (1,)), (8, 5, (64,)), (8, 2621440, (1000003,)),
                                            (7, 144779, (300, 257))])
def test_plan_matches_oracle(P, n_rows, shape, T):
    from hopwise_amd import ops, parallel as par

    rng = np.random.default_rng(P * 1000 + n_rows)
    ids = rng.integers(0, n_rows, shape)
    if ids.size > 10:
        ids.reshape(-1)[:5] = ids.reshape(-1)[5:10]  # repeats
    cap = par.capacity(ids.size, n_rows, P, 1.25)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev())
    inv, meta = ops.shard_plan(torch.from_numpy(ids).to(dev()), n_rows, P, cap, ovf, T)
    r_inv, r_meta, r_ovf = osp.plan(ids, n_rows, P, cap, T)
    assert r_ovf == 0 and int(ovf.item()) == 0
    assert np.array_equal(inv.cpu().numpy(), r_inv) and inv.shape == ids.shape
    meta = meta.cpu().numpy()
    assert np.array_equal(meta[:, 0], r_meta[:, 0])
    for o in range(P):  # slots past the count are unspecified
        c = r_meta[o, 0]
        assert np.array_equal(meta[o, 1 : 1 + c], r_meta[o, 1 : 1 + c])
    # the defining property: id = local * P + owner at the slot inv points to
    own, slot = r_inv.reshape(-1) // (T * cap), r_inv.reshape(-1) % (T * cap)
    assert np.array_equal(meta[own, 1 + slot].astype(np.int64) * P + own, ids.reshape(-1))


def test_plan_of_an_empty_lookup():
    from hopwise_amd import ops

    ovf = torch.zeros(1, dtype=torch.int32, device=dev())
    inv, meta = ops.shard_plan(torch.zeros(0, dtype=torch.int64, device=dev()), 100, 4, 3, ovf)
    assert inv.numel() == 0 and meta[:, 0].cpu().tolist() == [0, 0, 0, 0] and int(ovf.item()) == 0
    out = torch.full((12, 8), 7.0, device=dev())
    ops.shard_gather(torch.ones(25, 8, device=dev()), meta, 3, out)
    assert float(out.min()) == 7.0  # nothing requested: no row is written, none is read


def test_plan_overflow_flag_and_oob():
    from hopwise_amd import ops

    ids = np.arange(0, 64, dtype=np.int64) * 4  # all owned by rank 0 of 4
    ovf = torch.zeros(1, dtype=torch.int32, device=dev())
    inv, meta = ops.shard_plan(torch.from_numpy(ids).to(dev()), 1000, 4, 10, ovf)
    assert int(ovf.item()) == 64 and meta[:, 0].cpu().tolist() == [10, 0, 0, 0]
    assert int(inv.max().item()) < 4 * 10  # clamped into the compact table
    before = ops.oob_count(dev())
    bad = torch.tensor([5, -1, 2000], device=dev())
    ops.shard_plan(bad, 1000, 4, 8, torch.zeros(1, dtype=torch.int32, device=dev()))
    assert ops.oob_count(dev()) == before + 2


@pytest.mark.parametrize("P,dim,T", [(1, 64, 2), (2, 200, 2), (8, 256, 2), (3, 7, 1), (4, 32, 3)])
def test_virtual_ranks_exchange_roundtrip(P, dim, T):
    """Forward rows and backward row-gradients of T tables packed in one message per peer, through plan -> gather ->
    (hand-made all-to-all) -> scatter-add for P virtual ranks, each with its own ids, against dense indexing."""
    from hopwise_amd import ops, parallel as par

    n_rows, n = 777, 500
    rng = np.random.default_rng(P + dim)
    full = [rng.standard_normal((n_rows, dim)).astype(np.float32) for _ in range(T)]
    shards = [[torch.from_numpy(np.ascontiguousarray(f[r::P])).to(dev()) for f in full] for r in range(P)]
    ids = [rng.integers(0, n_rows, n) for _ in range(P)]
    cap = par.capacity(n, n_rows, P, 1.25)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev())
    plans = [ops.shard_plan(torch.from_numpy(i).to(dev()), n_rows, P, cap, ovf, T) for i in ids]
    assert int(ovf.item()) == 0
    # all-to-all by hand: what rank r receives from rank s is s's chunk for owner r
    recv_meta = [torch.stack([plans[s][1][r] for s in range(P)]).contiguous() for r in range(P)]
    sent = []
    for r in range(P):
        buf = torch.full((P * T * cap, dim), float("nan"), device=dev())
        for t in range(T):
            ops.shard_gather(shards[r][t], recv_meta[r], cap, buf, t, T)
        sent.append(buf.view(P, T * cap, dim))
    for s in range(P):
        back = torch.cat([sent[r][s] for r in range(P)])  # [P * T * cap, d] at requester s
        tabs = par._table_views(back, T, cap, dim)
        for t in range(T):
            assert np.array_equal(tabs[t][plans[s][0]].cpu().numpy(), full[t][ids[s]])
    # backward: requester s holds dense compact gradients per table; owner r adds what every s sends it
    rows = P * T * cap - (T - 1) * cap
    g = [[rng.standard_normal((rows, dim)).astype(np.float32) for _ in range(T)] for _ in range(P)]
    want = [np.zeros((n_rows, dim), np.float64) for _ in range(T)]
    packed = []
    for s in range(P):
        inv = plans[s][0].cpu().numpy()
        used = np.unique(inv)
        uniq_ids = np.zeros(rows, np.int64)
        uniq_ids[inv] = ids[s]
        send = torch.zeros((P, T, cap, dim), device=dev())
        for t in range(T):
            np.add.at(want[t], uniq_ids[used], g[s][t][used].astype(np.float64))
            gt = torch.from_numpy(g[s][t]).to(dev())
            send[:, t].copy_(gt.as_strided((P, cap, dim), (T * cap * dim, dim, 1)))
        packed.append(send.view(P, T * cap, dim))
    for r in range(P):
        back = torch.cat([packed[s][r] for s in range(P)]).contiguous()
        for t in range(T):
            acc = torch.zeros_like(shards[r][t])
            ops.shard_scatter_add_(acc, recv_meta[r], cap, back, t, T)
            np.testing.assert_allclose(acc.cpu().numpy(), want[t][r::P], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("model", ["RotatE", "ComplEx"])
def test_sharded_full_sort_single_rank_matches_plain(model):
    from hopwise_amd import parallel as par
    from tests.helpers import random_tables
    from tests.test_gpu_kge import build

    dim, nU, nI, nE, nR = 64, 40, 30, 90, 7
    tables = random_tables(model, nU, nE, nR, dim, seed=4)
    plain = build(model, tables, dim, nU, nI, nE, nR)
    sh = par.RowShardedKGE(build(model, tables, dim, nU, nI, nE, nR))
    users = torch.tensor([3, 1, 16, 3, 9], device=dev())
    hist = (torch.tensor([0, 0, 2, 4], device=dev()), torch.tensor([5, 2, 7, 1], device=dev()))
    with torch.no_grad():
        assert torch.equal(sh.full_sort_predict({"user_id": users}), plain.full_sort_predict({"user_id": users}))
        kg = {"head_id": torch.tensor([4, 28, 1], device=dev()), "relation_id": torch.tensor([1, 3, 0], device=dev())}
        assert torch.equal(sh.full_sort_predict_kg(kg), plain.full_sort_predict_kg(kg))
        for inter, h, is_kg in (({"user_id": users}, hist, False), (kg, None, True)):
            v0, i0 = plain.full_sort_topk(inter, 10, h, kg=is_kg)
            v1, i1 = sh.full_sort_topk(inter, 10, h, kg=is_kg)
            assert torch.equal(v0, v1) and torch.equal(i0, i1)
    sh.check_overflow()
