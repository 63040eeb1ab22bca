"""world_size-2 gloo (CPU) tests of the multi-GPU host logic: row-sharded lookup (forward + backward exchange),
the sharded KGE wrapper's id remapping and gradient routing, global top-k merge, max-over-ranks timing.
The device primitives (routing plan, row gather / scatter-add, full sort) are replaced by CPU stand-ins here
(parallel.set_shard_ops, oracle/shard_plan.py) -- only the exchange logic is under test; the HIP primitives have
their own GPU tests (tests/test_gpu_shard.py), which check them against the same oracle."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kge as okge


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_plan(ids, n_rows, P, cap, overflow, n_tables=1):
    from oracle import shard_plan as osp

    inv, meta, ovf = osp.plan(ids.numpy(), n_rows, P, cap, n_tables)
    if ovf:
        overflow[0] = max(int(overflow[0]), ovf)
    return torch.from_numpy(inv), torch.from_numpy(np.maximum(meta, 0))


def _cpu_gather(shard, meta, cap, out, table=0, n_tables=1):
    from oracle import shard_plan as osp

    # rows past a peer's count are garbage on the device: poison them here so that any use shows up
    out.view(meta.shape[0], n_tables, cap, -1)[:, table] = float("nan")
    osp.gather(shard.detach().numpy(), meta.numpy(), cap, out.numpy(), table, n_tables)
    return out


def _cpu_scatter_add(grad, meta, cap, src, table=0, n_tables=1):
    from oracle import shard_plan as osp

    osp.scatter_add(grad.numpy(), meta.numpy(), cap, src.numpy(), table, n_tables)
    return grad


def _rotate_scores(head, rel, tail, h_idx, r_idx, r_fixed, lo, hi, margin):
    """[B, hi - lo] RotatE scores of queries (h, r) against tail rows [lo, hi) -- torch CPU stand-in of the HIP full sort."""
    r = rel[0][r_idx] if r_idx is not None else rel[0][r_fixed].expand(len(h_idx), -1)
    rr, ri = torch.cos(r), torch.sin(r)
    hre, him = head[0][h_idx], head[1][h_idx]
    qre, qim = rr * hre - ri * him, rr * him + ri * hre
    re = qre[:, None, :] - tail[0][lo:hi][None]
    im = qim[:, None, :] - tail[1][lo:hi][None]
    return margin - torch.sqrt((re * re + im * im).sum(-1))


def _cpu_full_sort(model, head, rel, tail, h_idx, r_idx, r_fixed, lo, hi, margin=0.0):
    return _rotate_scores(head, rel, tail, h_idx, r_idx, r_fixed, lo, hi, margin).reshape(-1)


def _cpu_full_sort_topk(model, head, rel, tail, h_idx, r_idx, r_fixed, lo, hi, k, history_index=None, mask_first_col=True,
                        margin=0.0):
    sc = _rotate_scores(head, rel, tail, h_idx, r_idx, r_fixed, lo, hi, margin)
    if mask_first_col:
        sc[:, 0] = -np.inf
    if history_index is not None:
        hr, hc = history_index
        ok = (hc >= 0) & (hc < sc.shape[1])
        sc[hr[ok], hc[ok]] = -np.inf
    return _cpu_topk(sc, k)


def _cpu_topk(v, k):
    order = np.lexsort((np.broadcast_to(np.arange(v.shape[1]), v.shape), -v.numpy().astype(np.float64)), axis=1)[:, :k]
    order = torch.from_numpy(order.copy())
    return torch.gather(v, 1, order), order


class _StubModel(torch.nn.Module):
    """Same table/field layout as hopwise_amd.model.RotatE, with a torch-CPU calculate_loss standing in for the
    HIP scorer (the kernels cannot run in this container)."""

    USER_TABLES = ("user_embedding", "user_embedding_im")
    ENTITY_TABLES = ("entity_embedding", "entity_embedding_im")
    RELATION_TABLES = ("relation_embedding",)
    MODEL, margin = "RotatE", 1.0
    USER_ID, ITEM_ID, NEG_ITEM_ID = "user_id", "item_id", "neg_item_id"
    HEAD_ENTITY_ID, RELATION_ID, TAIL_ENTITY_ID, NEG_TAIL_ENTITY_ID = "head_id", "relation_id", "tail_id", "neg_tail_id"

    def __init__(self, tables, n_users, n_entities, n_items=None):
        super().__init__()
        self.n_users, self.n_entities, self.n_items = n_users, n_entities, n_items
        for k, v in tables.items():
            emb = torch.nn.Embedding(*v.shape)
            emb.weight.data.copy_(torch.from_numpy(v))
            setattr(self, k, emb)

    def _w(self, names):
        return tuple(getattr(self, n).weight for n in names)

    def _rec_rel_full_sort(self):
        return self.relation_embedding.weight.shape[0] - 1

    def _tables(self, side):
        head = self._w(self.USER_TABLES if side == "rec" else self.ENTITY_TABLES)
        return head, self._w(self.RELATION_TABLES), self._w(self.ENTITY_TABLES)

    def calculate_loss(self, it):
        def score(side, h, r, t):
            H, R, T = self._tables(side)
            rr, ri = torch.cos(R[0][r]), torch.sin(R[0][r])
            re = rr * H[0][h] - ri * H[1][h] - T[0][t]
            im = rr * H[1][h] + ri * H[0][h] - T[1][t]
            return 1.0 - torch.sqrt((re * re + im * im).sum(-1))

        ui = torch.full_like(it["user_id"], self.relation_embedding.weight.shape[0] - 1)
        bce = torch.nn.BCEWithLogitsLoss()
        sp, sn = score("rec", it["user_id"], ui, it["item_id"]), score("rec", it["user_id"], ui, it["neg_item_id"])
        kp, kn = score("kg", it["head_id"], it["relation_id"], it["tail_id"]), score("kg", it["head_id"], it["relation_id"], it["neg_tail_id"])
        lab = lambda a, b: torch.cat([torch.ones_like(a), torch.zeros_like(b)])
        return bce(torch.cat([sp, sn]), lab(sp, sn)) + bce(torch.cat([kp, kn]), lab(kp, kn))


def _worker(rank, world_size, port, tmp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    torch.set_num_threads(1)
    from hopwise_amd import parallel as par

    par.set_shard_ops(plan=_cpu_plan, gather=_cpu_gather, scatter_add=_cpu_scatter_add, full_sort=_cpu_full_sort,
                      full_sort_topk=_cpu_full_sort_topk, topk=_cpu_topk)
    try:
        # ---- 1. lookup forward / backward against a dense table -----------------------------------------
        rng = np.random.default_rng(0)  # same on every rank
        N, d = 53, 6
        full_a = torch.from_numpy(rng.standard_normal((N, d)).astype(np.float32))
        full_b = torch.from_numpy(rng.standard_normal((N, d)).astype(np.float32))
        sa = par.shard_rows(full_a, rank, world_size).requires_grad_(True)
        sb = par.shard_rows(full_b, rank, world_size).requires_grad_(True)
        ids = torch.from_numpy(np.random.default_rng(10 + rank).integers(0, N, (7, 5)))  # differs per rank, repeats
        (ca, cb), inv, ovf = par.sharded_lookup((sa, sb), ids, N)
        assert int(ovf.item()) == 0
        assert torch.equal(ca[inv], full_a[ids]) and torch.equal(cb[inv], full_b[ids])
        cap_ = par.capacity(ids.numel(), N, world_size, 1.25)
        assert ca.shape[0] == cb.shape[0] == world_size * 2 * cap_ - cap_  # two tables in one fixed-size message
        w = torch.from_numpy(np.random.default_rng(20 + rank).standard_normal((7, 5, d)).astype(np.float32))
        ((ca[inv] * w).sum() + 2.0 * (cb[inv] * w).sum()).backward()
        # dense expectation: every rank's contribution, summed at the owner
        all_ids = [torch.from_numpy(np.random.default_rng(10 + r).integers(0, N, (7, 5))) for r in range(world_size)]
        all_w = [torch.from_numpy(np.random.default_rng(20 + r).standard_normal((7, 5, d)).astype(np.float32)) for r in range(world_size)]
        dense = torch.zeros(N, d)
        for i_, w_ in zip(all_ids, all_w):
            dense.index_add_(0, i_.reshape(-1), w_.reshape(-1, d))
        assert torch.allclose(sa.grad, dense[rank::world_size], atol=1e-6)
        assert torch.allclose(sb.grad, 2 * dense[rank::world_size], atol=1e-6)

        # ---- 2. sharded KGE wrapper: loss and gradients equal the unsharded model on the same batch ------
        nU, nI, nE, nR, dim = 17, 11, 29, 5, 8
        rng = np.random.default_rng(3)
        tables = {k: rng.standard_normal(((nU if k.startswith("user") else nR if k.startswith("rel") else nE), dim)).astype(np.float32) * 0.3
                  for k in okge.table_names("RotatE")}
        def batch(r):
            g = np.random.default_rng(100 + r)
            return {k: torch.from_numpy(v) for k, v in dict(
                user_id=g.integers(1, nU, 9), item_id=g.integers(1, nI, 9), neg_item_id=g.integers(1, nI, 9),
                head_id=g.integers(1, nE, 12), relation_id=g.integers(1, nR - 1, 12), tail_id=g.integers(1, nE, 12),
                neg_tail_id=g.integers(1, nE, 12)).items()}
        ref = _StubModel(tables, nU, nE)
        ref_loss = sum(ref.calculate_loss(batch(r)) for r in range(world_size)) / world_size
        ref_loss.backward()
        sharded = par.RowShardedKGE(_StubModel(tables, nU, nE, nI))
        loss = sharded.calculate_loss(batch(rank))
        loss.backward()
        sharded.sync_grads()
        sharded.check_overflow()
        for n in _StubModel.USER_TABLES + _StubModel.ENTITY_TABLES:
            want = getattr(ref, n).weight.grad[rank::world_size]
            assert torch.allclose(getattr(sharded, "shard_" + n).grad, want, atol=1e-6), n
        assert torch.allclose(sharded.model.relation_embedding.weight.grad, ref.relation_embedding.weight.grad, atol=1e-6)
        lt = loss.detach().clone()
        dist.all_reduce(lt)
        assert abs(lt.item() / world_size - ref_loss.item()) < 1e-6
        sd = sharded.full_state_dict()
        for k, v in tables.items():
            assert torch.equal(sd[k + ".weight"], torch.from_numpy(v)), k

        # ---- 2b. sharded full sort == unsharded full sort, values and ranked ids exact ---------------
        full = _StubModel(tables, nU, nE, nI)
        users = torch.tensor([3, 1, 16, 3, 9])
        hr, hc = torch.tensor([0, 0, 2, 4, 4, 1]), torch.tensor([5, 2, 7, 1, 10, 3])
        for kg, inter, n_cols in ((False, {"user_id": users}, nI),
                                  (True, {"head_id": torch.tensor([4, 28, 1, 17]), "relation_id": torch.tensor([1, 3, 0, 2])}, nE)):
            H, R, T = full._tables("kg" if kg else "rec")
            q = inter["head_id"] if kg else users
            want = _rotate_scores(H, R, T, q, inter.get("relation_id"), full._rec_rel_full_sort(), 0, n_cols, 1.0)
            got = (sharded.full_sort_predict_kg if kg else sharded.full_sort_predict)(inter).view(-1, n_cols)
            assert torch.equal(got, want), kg
            hist = None if kg else (hr, hc)
            masked = want.clone()
            masked[:, 0] = -np.inf
            if hist is not None:
                masked[hist] = -np.inf
            for k in (3, 10):
                rv, ri = okge.topk_desc(masked.detach().numpy(), k)
                tv, ti = sharded.full_sort_topk(inter, k, hist, kg=kg)
                assert np.array_equal(ti.numpy(), ri) and np.array_equal(tv.numpy(), rv), (kg, k)
        sharded.check_overflow()

        # ---- 2c. capacity overflow is detected, not silently wrong -------------------------------------------------
        tight = par.RowShardedKGE(_StubModel(tables, nU, nE, nI), capacity_factor=0.01)
        tight.capacity_factor = 0.0
        _orig_cap = par.capacity
        par.capacity = lambda n, rows, P, f: 2  # two distinct rows per owner: the 12-triple batch cannot fit
        try:
            tight.calculate_loss(batch(rank))
            with pytest.raises(RuntimeError, match="capacity"):
                tight.check_overflow()
        finally:
            par.capacity = _orig_cap

        # ---- 2d. overflow never reaches the optimizer: every id owned by rank 0 (id % P == 0), capacity sized for a spread ----
        skew = par.RowShardedKGE(_StubModel(tables, nU, nE, nI), capacity_factor=1.0)
        _orig_cap = par.capacity
        par.capacity = lambda n, rows, P, f: 3
        try:
            b0 = batch(rank)
            for k in ("item_id", "neg_item_id", "head_id", "tail_id", "neg_tail_id"):
                b0[k] = (b0[k] // world_size) * world_size  # all owned by rank 0
            b0["item_id"] = b0["item_id"].clamp(max=((nI - 1) // world_size) * world_size)
            loss = skew.calculate_loss(b0)
            assert torch.isnan(loss).item()  # decided by the device flag: the trainer's NaN check stops here
            loss.backward()
            with pytest.raises(RuntimeError, match="capacity"):
                skew.sync_grads()  # ... and a caller that ignores the loss is stopped before its optimizer step
        finally:
            par.capacity = _orig_cap

        # ---- 2f. only ONE rank overflows: the decision is still collective -- NaN loss on both ranks, both stopped before the
        #          gradient all-reduce, nobody left waiting in a collective; the flag is reported once and the wrapper goes on ----
        lone = par.RowShardedKGE(_StubModel(tables, nU, nE, nI), capacity_factor=1.0)
        par.capacity = lambda n, rows, P, f: 5
        try:
            b1 = batch(rank)
            if rank == 0:  # a narrow batch: ids 1..4, at most two distinct rows per owner
                b1 = {k: v % 4 + 1 for k, v in b1.items()}
            mine = torch.zeros(1, dtype=torch.int32)
            _cpu_plan(torch.cat([b1[k] for k in ("item_id", "neg_item_id", "head_id", "tail_id", "neg_tail_id")]), nE,
                      world_size, 5, mine)
            assert (int(mine.item()) > 0) == (rank == 1)  # this rank's own routing: only rank 1 does not fit
            loss = lone.calculate_loss(b1)
            assert torch.isnan(loss).item()  # ... yet BOTH ranks see the NaN the trainer's check stops at
            with pytest.raises(RuntimeError, match="capacity"):
                lone.sync_grads()  # and both are stopped here, ahead of the all-reduce
        finally:
            par.capacity = _orig_cap
        loss = lone.calculate_loss(batch(rank))  # reported once: at the real capacity the same wrapper goes on
        assert torch.isfinite(loss).item()
        loss.backward()
        lone.sync_grads()

        # ---- 2e. ranks with different batch sizes (a ragged last batch): the message sizes come from max_ids, not from the
        #          local batch, so the equal-split all-to-alls still line up -----------------------------------------------
        ragged = par.RowShardedKGE(_StubModel(tables, nU, nE, nI), max_ids={"train_entity": 5 * 12, "train_user": 9})
        def cut(b, n_rec, n_kg):
            return {k: (v[:n_rec] if k in ("user_id", "item_id", "neg_item_id") else v[:n_kg]) for k, v in b.items()}
        sizes = [(9 - 2 * r, 12 - 3 * r) for r in range(world_size)]
        ref2 = _StubModel(tables, nU, nE)
        ref2_loss = sum(ref2.calculate_loss(cut(batch(r), *sizes[r])) for r in range(world_size)) / world_size
        ref2_loss.backward()
        l2 = ragged.calculate_loss(cut(batch(rank), *sizes[rank]))
        l2.backward()
        ragged.sync_grads()
        for n in _StubModel.USER_TABLES + _StubModel.ENTITY_TABLES:
            assert torch.allclose(getattr(ragged, "shard_" + n).grad, getattr(ref2, n).weight.grad[rank::world_size], atol=1e-6), n
        assert torch.allclose(ragged.model.relation_embedding.weight.grad, ref2.relation_embedding.weight.grad, atol=1e-6)
        # a batch LARGER than max_ids is not refused by one rank on its own (its peers would wait in the all-to-all): it is
        # routed like any other, and fits here (the capacity is the whole shard)
        (ca2, cb2), inv2, ovf2 = par.sharded_lookup((sa.detach(), sb.detach()), ids, N, max_ids=ids.numel() - 1)
        assert int(ovf2.item()) == 0 and torch.equal(ca2[inv2], full_a[ids]) and torch.equal(cb2[inv2], full_b[ids])
        # without max_ids the first batch freezes the maximum over ranks (rank 0 has the larger batch here)
        auto = par.RowShardedKGE(_StubModel(tables, nU, nE, nI))
        auto.calculate_loss(cut(batch(rank), *sizes[rank]))
        assert auto._max_ids["train_user"] == 9 and auto._max_ids["train_entity"] == 2 * 9 + 3 * 12

        # ---- 3. global top-k merge ---------------------------------------------------------------------------
        rng = np.random.default_rng(5)
        scores = rng.standard_normal((4, 40)).astype(np.float32)
        scores[1, 7] = scores[1, 8]  # a cross-rank tie: smaller global id must win
        mine = np.arange(rank, 40, world_size)
        lv, lo = _cpu_topk(torch.from_numpy(scores[:, mine].copy()), 6)
        lids = torch.from_numpy(mine)[lo]
        tv, ti = par.merge_topk(lv, lids, 6, topk_fn=_cpu_topk)
        rv, ri = okge.topk_desc(scores, 6)
        assert np.array_equal(ti.numpy(), ri) and np.array_equal(tv.numpy(), rv)

        # ---- 4. replicas (the PGPR searches): round-robin user partition, per-user rows back in the original order ----
        users = torch.arange(100, 100 + 37)
        for n_total in (37, 36, 1):
            u = users[:n_total]
            mine_u = par.replica_users(u)
            assert mine_u.tolist() == u.tolist()[rank::world_size]
            res = torch.stack([mine_u * 3, mine_u + 7], dim=1)  # any per-user rows
            got = par.gather_user_results(res, n_total)
            assert got.tolist() == torch.stack([u * 3, u + 7], dim=1).tolist()

        # ---- 5. timing helper -----------------------------------------------------------------------------------
        assert par.max_over_ranks(1.0 + rank, torch.device("cpu")) == float(world_size)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        par.set_shard_ops()
        dist.destroy_process_group()


def _worker8(rank, world_size, port, tmp):
    """The control flow of `bench.py --gpus 8 --workload sharded` before it meets RCCL: eight ranks, Yelp2018 row counts,
    Zipf-skewed ids (hot entities), the per-rank batch of the bench leg -- loss and every gradient shard against the
    unsharded model, no overflow at the default capacity factor, sharded full sort exact."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    torch.set_num_threads(1)
    from hopwise_amd import parallel as par

    par.set_shard_ops(plan=_cpu_plan, gather=_cpu_gather, scatter_add=_cpu_scatter_add, full_sort=_cpu_full_sort,
                      full_sort_topk=_cpu_full_sort_topk, topk=_cpu_topk)
    try:
        nU, nI, nE, nR, dim, B = 45920, 45539, 90962, 44, 4, 512
        rng = np.random.default_rng(7)
        tables = {k: rng.standard_normal(((nU if k.startswith("user") else nR if k.startswith("rel") else nE), dim)).astype(np.float32) * 0.3
                  for k in okge.table_names("RotatE")}

        def zipf(g, n, size):  # rank-frequency ~ 1 / rank over [1, n)
            return np.minimum((g.zipf(1.3, size) - 1) % (n - 1) + 1, n - 1)

        def batch(r):
            g = np.random.default_rng(500 + r)
            return {k: torch.from_numpy(v.astype(np.int64)) for k, v in dict(
                user_id=zipf(g, nU, B), item_id=zipf(g, nI, B), neg_item_id=g.integers(1, nI, B),
                head_id=zipf(g, nE, B), relation_id=g.integers(1, nR - 1, B), tail_id=zipf(g, nE, B),
                neg_tail_id=g.integers(1, nE, B)).items()}

        ref = _StubModel(tables, nU, nE)
        ref_loss = sum(ref.calculate_loss(batch(r)) for r in range(world_size)) / world_size
        ref_loss.backward()
        sharded = par.RowShardedKGE(_StubModel(tables, nU, nE, nI), max_ids={"train_entity": 5 * B, "train_user": B})
        loss = sharded.calculate_loss(batch(rank))
        assert torch.isfinite(loss).item()
        loss.backward()
        sharded.sync_grads()  # raises on overflow
        for n in _StubModel.USER_TABLES + _StubModel.ENTITY_TABLES:
            want = getattr(ref, n).weight.grad[rank::world_size]
            assert torch.allclose(getattr(sharded, "shard_" + n).grad, want, atol=1e-5), n
        assert torch.allclose(sharded.model.relation_embedding.weight.grad, ref.relation_embedding.weight.grad, atol=1e-5)
        lt = loss.detach().clone()
        dist.all_reduce(lt)
        assert abs(lt.item() / world_size - ref_loss.item()) < 1e-5
        # evaluation: 6 users against all 45 539 items, top-10 equal to the unsharded ranking
        full = _StubModel(tables, nU, nE, nI)
        users = torch.tensor([3, 1, 45919, 3, 9, 20000])
        H, R, T = full._tables("rec")
        want = _rotate_scores(H, R, T, users, None, full._rec_rel_full_sort(), 0, nI, 1.0)
        want[:, 0] = -np.inf
        rv, ri = okge.topk_desc(want.detach().numpy(), 10)
        tv, ti = sharded.full_sort_topk({"user_id": users}, 10, None, kg=False)
        assert np.array_equal(ti.numpy(), ri) and np.array_equal(tv.numpy(), rv)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        par.set_shard_ops()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world_size_8_gloo_yelp_shape_zipf(tmp_path):
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(8))


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
