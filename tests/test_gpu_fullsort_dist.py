"""The full sort of the distance scorers (TransE: -||h + r - t||, transe.py:55-57, 112-154; hopwise's RotatE: margin - the L2
norm over the 2 d (re, im) differences, rotate.py:61-69, 161-220) as ONE product on the MFMA GEMM + the finish pass
(csrc/kge_fullsort_gemm.hip), against the gather kernel (itself pinned by the reference goldens in test_gpu_kge.py) and the
numpy oracle; and the evaluation top-k of that form, whose exact re-rank must return what the gather path returns."""

import numpy as np
import pytest
import torch

from oracle import kge as okge
from tests.helpers import random_tables
from tests.test_gpu_kge import build, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _default_routes(monkeypatch):
    """The tests of this module name the route they exercise by config key (or set the switch themselves): a caller's
    HW_* route switches must not turn the default product routes off underneath them."""
    for e in ("HW_FULLSORT_RANK_PLANES", "HW_FULLSORT_RANK_TERMS", "HW_GEMM_SPLIT"):
        monkeypatch.delenv(e, raising=False)

SHAPES = ((64, 1003, 700, 3, 10), (200, 777, 300, 17, 16), (24, 200, 130, 130, 20), (8, 90, 40, 1, 5), (30, 411, 97, 9, 10))


def _inputs(model, dim, nE, nI, B, nU=50, nR=6, near=True):
    tables = random_tables(model, nU, nE, nR, dim, seed=dim + B)
    rng = np.random.default_rng(B)
    users_np, heads_np, rels_np = rng.integers(0, nU, B), rng.integers(0, nE, B), rng.integers(0, nR, B)
    if near:
        # trained embeddings put a user's items NEXT to its query: plant candidates at the query itself and at 1e-3 / 1e-2 / 0.1 of
        # its norm away from it -- where ||q||^2 + ||t||^2 - 2 <q, t> cancels and the finish pass must recompute from the rows
        hn, rn, tn = okge.slots(model, "rec")
        H = [tables[n][users_np] for n in hn]
        r = tables[rn[0]][-1]
        if model == "TransE":
            q = H[0] + r
        else:
            q = np.concatenate([np.cos(r) * H[0] - np.sin(r) * H[1], np.cos(r) * H[1] + np.sin(r) * H[0]], 1)
        d = tables[tn[0]].shape[1]
        for b in range(min(B, 4)):
            for j, eps in enumerate((0.0, 1e-3, 1e-2, 0.1)):
                item = 1 + (7 * b + j) % (nI - 1)
                noise = rng.standard_normal(q.shape[1]).astype(np.float32)
                row = (q[b] + eps * np.linalg.norm(q[b]) * noise / np.linalg.norm(noise)).astype(np.float32)
                for k, nm in enumerate(tn):
                    tables[nm][item] = row[k * d : (k + 1) * d]
    return tables, users_np, heads_np, rels_np


@pytest.mark.parametrize("model", ["TransE", "RotatE"])
def test_gemm_full_sort_equals_the_gather_kernel_and_the_oracle(model):
    for dim, nE, nI, B, k in SHAPES:
        nU, nR = 50, 6
        tables, users_np, heads_np, rels_np = _inputs(model, dim, nE, nI, B)
        m = build(model, tables, dim, nU, nI, nE, nR, gemm_min_scores=0)
        assert m.GEMM_FORM == "distance" and m.gemm_full_sort and m.gemm_rerank
        users, heads, rels = (torch.from_numpy(x).to(dev()) for x in (users_np, heads_np, rels_np))
        with torch.no_grad():
            for kg, inter, n in ((False, {"user_id": users}, nI), (True, {"head_id": heads, "relation_id": rels}, nE)):
                fn = m.full_sort_predict_kg if kg else m.full_sort_predict
                assert m._use_gemm_full_sort(B)
                got = fn(inter).view(B, n)
                assert got.is_contiguous()  # what hopwise's trainer views and masks in place (trainer.py:731-734)
                m.gemm_full_sort = False
                ker = fn(inter).view(B, n)
                m.gemm_full_sort = True
                want = (okge.full_sort_predict_kg(model, tables, heads_np, rels_np, m.margin) if kg else
                        okge.full_sort_predict(model, tables, users_np, nI, m.margin, m._rec_rel_full_sort()))
                # against the gather kernel: the distance to fp32 rounding (1e-5 of itself + of the rows' scale)
                dist_k = (m.margin if model == "RotatE" else 0.0) - ker.cpu().numpy()
                dist_g = (m.margin if model == "RotatE" else 0.0) - got.cpu().numpy()
                scale = float(np.abs(dist_k).max())
                np.testing.assert_allclose(dist_g, dist_k, rtol=2e-5, atol=2e-6 * scale)
                np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("model", ["TransE", "RotatE"])
def test_finish_pass_recomputes_where_the_expansion_cancels(model):
    """ops level: candidates planted on and next to the queries.  Without the recomputation (rescore_frac = 0) the planted
    entries are off by far more than the tolerance; with it every entry is within 1e-5 relative of a float64 distance."""
    from hopwise_amd import ops

    dim, nU, nE, nI, nR, B = 200, 40, 600, 500, 5, 8
    tables = random_tables(model, nU, nE, nR, dim, seed=3)
    m = build(model, tables, dim, nU, nI, nE, nR)
    users = torch.arange(1, 1 + B, device=dev())
    head, rel, tail = m._tables("rec")
    with torch.no_grad():
        q, qn = ops.kge_query_rows(model, head, rel, tail, users, None, m._rec_rel_full_sort(), sqnorm=True, margin=m.margin)
        K = q.shape[1]
        table = m._cat_tables(tail)[:nI].clone().contiguous()
        rng = np.random.default_rng(0)
        planted = []
        for b in range(B):
            for j, eps in enumerate((0.0, 1e-4, 1e-3, 1e-2, 0.05, 0.2)):
                item = 1 + 6 * b + j
                noise = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).to(dev())
                table[item] = q[b] + eps * q[b].norm() * noise / noise.norm()
                planted.append((b, item))
        tn = ops.rows_sqnorm(table)
        np.testing.assert_allclose(tn.cpu().numpy(), (table.double() ** 2).sum(1).cpu().numpy(), rtol=2e-6)
        np.testing.assert_allclose(qn.cpu().numpy(), (q.double() ** 2).sum(1).cpu().numpy(), rtol=2e-6)
        want = torch.cdist(q.double(), table.double())  # float64 distances
        for padded in (False, True):
            def dots():
                out = None
                if padded:
                    out = torch.empty((B, (nI + 63) // 64 * 64), dtype=torch.float32, device=dev())[:, :nI]
                return ops.gemm(q, table, out=out)

            counter = torch.zeros(1, dtype=torch.int32, device=dev())
            fixed = ops.kge_dist_finish_(dots(), q, table, qn, tn, base=0.0, counter=counter)
            raw = ops.kge_dist_finish_(dots(), q, table, qn, tn, base=0.0, rescore_frac=0.0)
            assert int(counter.item()) >= 4 * B  # the planted entries up to 5 % away, at least
            err_fixed = ((-fixed.double()) - want).abs()
            err_raw = ((-raw.double()) - want).abs()
            tol = 1e-5 * want + 1e-6 * float(want.max())
            assert bool((err_fixed <= tol).all()), float((err_fixed - tol).max())
            rows, cols = zip(*planted)
            assert float(err_raw[list(rows), list(cols)].max()) > 50 * float(err_fixed[list(rows), list(cols)].max() + 1e-9)
            # the query itself: distance exactly 0
            assert float(fixed[0, 1]) == 0.0
            # the fused form (hw_gemm_dist: the same recomputation inside the product's epilogue) on the same inputs
            out = torch.empty((B, (nI + 63) // 64 * 64), dtype=torch.float32, device=dev())[:, :nI] if padded else None
            fused = ops.gemm_dist(q, table, qn, tn, base=0.0, out=out)
            err_fused = ((-fused.double()) - want).abs()
            assert bool((err_fused <= tol).all()), float((err_fused - tol).max())
            assert float(fused[0, 1]) == 0.0
        # column 0 masked in the same pass
        masked = ops.kge_dist_finish_(ops.gemm(q, table), q, table, qn, tn, base=m.margin, mask_column0=True)
        assert bool(torch.isinf(masked[:, 0]).all()) and bool(torch.isfinite(masked[:, 1:]).all())


@pytest.mark.parametrize("model", ["TransE", "RotatE"])
def test_evaluation_topk_of_the_product_form_is_the_gather_paths(model):
    """full_sort_topk through product -> mask -> top-16 -> exact re-rank against the fused gather kernel: same columns in
    the same order; the values are predict()'s, bit for bit."""
    from hopwise_amd import ops

    for dim, nE, nI, B, k in SHAPES + ((64, 9000, 8600, 33, 10),):
        nU, nR = 50, 6
        tables, users_np, heads_np, rels_np = _inputs(model, dim, nE, nI, B)
        m = build(model, tables, dim, nU, nI, nE, nR, gemm_min_scores=0)
        rng = np.random.default_rng(dim)
        users, heads, rels = (torch.from_numpy(x).to(dev()) for x in (users_np, heads_np, rels_np))
        with torch.no_grad():
            for kg, inter, n in ((False, {"user_id": users}, nI), (True, {"head_id": heads, "relation_id": rels}, nE)):
                hist = (torch.from_numpy(rng.integers(0, B, 60)).to(dev()), torch.from_numpy(rng.integers(0, n, 60)).to(dev()))
                kk = min(k, n)
                gv, gi = m.full_sort_topk(inter, kk, hist, kg=kg)
                m.gemm_full_sort = False
                kv, ki = m.full_sort_topk(inter, kk, hist, kg=kg)
                m.gemm_full_sort = True
                kvn, kin, gvn, gin = kv.cpu().numpy(), ki.cpu().numpy(), gv.cpu().numpy(), gi.cpu().numpy()
                np.testing.assert_allclose(gvn, kvn, rtol=1e-6, atol=1e-6)
                # the two exact kernels sum a row in different orders (one ulp): positions may swap only inside such a tie
                gap_ok = np.ones_like(kin, bool)
                gaps = kvn[:, :-1] - kvn[:, 1:]
                gap_ok[:, :-1] &= gaps > 4e-7 * np.abs(kvn[:, :-1]) + 1e-7
                gap_ok[:, 1:] &= gaps > 4e-7 * np.abs(kvn[:, :-1]) + 1e-7
                assert (kin == gin)[gap_ok].all()
                assert gap_ok.mean() > 0.8
                # values == predict() on the returned (query, column) pairs
                if not kg:
                    flat = {"user_id": users.repeat_interleave(kk), "item_id": gi.reshape(-1)}
                    np.testing.assert_array_equal(m.predict(flat).view(B, kk).cpu().numpy(), gvn)
                # history and column 0 never come back
                assert not (gin == 0).any()
                hr, hc = hist[0].cpu().numpy(), hist[1].cpu().numpy()
                banned = set(zip(hr.tolist(), hc.tolist()))
                assert not any((r, int(c)) in banned for r in range(B) for c in gin[r])


def test_rerank_raises_its_flag_and_the_exact_kernel_takes_over():
    """All candidates equal (a model before training, zero-initialised rows): every approximate score ties, the re-rank cannot
    prove its 16 candidates suffice, and the predicated gather kernel behind it must produce the result -- the gather path's,
    bit for bit."""
    from hopwise_amd import ops

    model, dim, nU, nE, nI, nR, B, k = "TransE", 64, 20, 400, 300, 4, 6, 10
    tables = random_tables(model, nU, nE, nR, dim, seed=0)
    tables["entity_embedding"][:] = tables["entity_embedding"][5]
    m = build(model, tables, dim, nU, nI, nE, nR, gemm_min_scores=0)
    inter = {"user_id": torch.arange(1, 1 + B, device=dev())}
    with torch.no_grad():
        gv, gi = m.full_sort_topk(inter, k)
        m.gemm_full_sort = False
        kv, ki = m.full_sort_topk(inter, k)
    assert torch.equal(gi, ki) and torch.equal(gv, kv)
    assert gi[0].tolist() == list(range(1, 1 + k))  # (value desc, column asc) among equals, column 0 masked

    # the flag logic by itself
    idx = torch.arange(40, 72, device=dev()).view(1, 32).repeat(3, 1)
    exact = torch.linspace(1.0, 0.0, 32, device=dev()).view(1, 32).repeat(3, 1).contiguous()
    approx = exact.clone()
    approx[1] = exact[1, k - 1] + 1.0      # row 1: the candidates' smallest approximate score is ABOVE the exact k-th value
    approx[2, 3] = float("-inf")           # row 2: a masked column among the candidates -> the set holds every live column
    val, out_idx, flag = ops.topk_rerank(idx, approx, exact, k)
    assert int(flag.item()) == 1
    val2, _, flag2 = ops.topk_rerank(idx[[0, 2]], approx[[0, 2]], exact[[0, 2]], k)
    assert int(flag2.item()) == 0
    assert val2[0].tolist() == exact[0, :k].tolist()
    assert out_idx[2].tolist() == [40, 41, 42] + list(range(44, 44 + k - 3))  # the masked candidate dropped out
    nan = exact.clone()
    nan[0, 0] = float("nan")
    assert int(ops.topk_rerank(idx, approx, nan, k)[2].item()) == 1
    # the distance form: approximate values are p = <q, t> - ||t||^2 / 2 of the biased product, the bound is taken on
    # d^2 = ||q||^2 - 2 p.  Distances 1, 2, ..., 32 (scores -1 ... -32), ||q||^2 = 4:
    dist = torch.arange(1, 33, device=dev(), dtype=torch.float32).view(1, 32)
    qn = torch.tensor([4.0], device=dev())
    p = (qn.view(1, 1) - dist * dist) / 2
    # k = 10: the k-th exact score is -10, a column outside the set is at distance >= 32 -- proven
    assert int(ops.topk_rerank(idx[:1], p, -dist, 10, rel_slack=1e-5, qn=qn, base=0.0, tn_max=900.0)[2].item()) == 0
    # the set's smallest p belongs to a column whose distance is no larger than the k-th one: not proven
    p_bad = p.clamp_min((4.0 - 9.5 * 9.5) / 2)  # every candidate looked at least as near as 9.5: so may a column outside
    assert int(ops.topk_rerank(idx[:1], p_bad, -dist, 10, rel_slack=1e-5, qn=qn, base=0.0, tn_max=900.0)[2].item()) == 1
    # with a margin as base (RotatE) the same decision
    assert int(ops.topk_rerank(idx[:1], p, 6.0 - dist, 10, rel_slack=1e-5, qn=qn, base=6.0, tn_max=900.0)[2].item()) == 0


def test_query_rows_kernel_equals_the_torch_fold():
    """hw_kge_query_rows against the folds written out in torch (kge_model.h: Scorer::fold), all five product-form models,
    vector and scalar row shapes, a fixed relation and per-row relations."""
    from hopwise_amd import ops

    for model in ops.GEMM_QUERY_MODELS:
        for dim in (8, 30, 64, 200, 260):
            nU, nE, nI, nR, B = 33, 70, 50, 7, 21
            tables = random_tables(model, nU, nE, nR, dim, seed=dim)
            m = build(model, tables, dim, nU, nI, nE, nR)
            rng = np.random.default_rng(dim)
            for side in ("rec", "kg"):
                head, rel, tail = m._tables(side)
                h_idx = torch.from_numpy(rng.integers(0, head[0].shape[0], B)).to(dev())
                r_idx = torch.from_numpy(rng.integers(0, nR, B)).to(dev()) if side == "kg" else None
                r_fixed = 0 if side == "kg" else m._rec_rel_full_sort()
                with torch.no_grad():
                    q, qn = ops.kge_query_rows(model, head, rel, tail, h_idx, r_idx, r_fixed, sqnorm=True, margin=m.margin)
                    h = [t.detach()[h_idx].double() for t in head]
                    r = [(t.detach()[r_idx] if r_idx is not None else t.detach()[r_fixed].unsqueeze(0)).double() for t in rel]
                    if model == "TransE":
                        want = h[0] + r[0]
                    elif model == "DistMult":
                        want = h[0] * r[0]
                    elif model == "RotatE":
                        cs, sn = torch.cos(r[0]), torch.sin(r[0])
                        want = torch.cat([cs * h[0] - sn * h[1], cs * h[1] + sn * h[0]], 1)
                    elif model == "ComplEx":
                        want = torch.cat([h[0] * r[0], h[1] * r[0] + h[0] * r[1] - h[1] * r[1]], 1)
                    else:
                        want = torch.cat([h[0] * r[0], h[1] * r[1] - h[2] * r[2], h[1] * r[2] + h[2] * r[1]], 1)
                    assert q.shape == want.shape
                    np.testing.assert_allclose(q.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-6)
                    np.testing.assert_allclose(qn.cpu().numpy(), (want ** 2).sum(1).cpu().numpy(), rtol=1e-5)


def test_cached_tables_and_norms_follow_the_weights():
    """RotatE keeps a cached [N, 2 d] copy of (re | im) and its row norms for the product: rebuilt after an optimizer step."""
    from hopwise_amd.optim import FusedAdam

    dim, nU, nI, nE, nR = 16, 30, 40, 90, 5
    m = build("RotatE", random_tables("RotatE", nU, nE, nR, dim, seed=1), dim, nU, nI, nE, nR, gemm_min_scores=0)
    inter = {"user_id": torch.arange(1, 9, device=dev())}
    with torch.no_grad():
        a = m.full_sort_predict(inter).clone()
    opt = FusedAdam(m.parameters(), lr=0.05)
    g = torch.Generator().manual_seed(0)
    for p in m.parameters():
        p.grad.copy_(torch.randn(p.shape, generator=g).to(p.device))
    opt.mark_touched()
    opt.step()
    with torch.no_grad():
        b = m.full_sort_predict(inter)
        m.gemm_full_sort = False
        c = m.full_sort_predict(inter).view_as(b)
    assert float((a - b).abs().max()) > 1e-3
    np.testing.assert_allclose(b.cpu().numpy(), c.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(c.abs().max()))


@pytest.mark.parametrize("path", [p for p in __import__("tests.test_gpu_kge", fromlist=["KGE_FILES"]).KGE_FILES
                                  if "kge_TransE_" in p or "kge_RotatE_" in p],
                         ids=lambda p: __import__("os").path.basename(p)[4:-4])
def test_product_form_against_the_reference_goldens(path):
    """The reference's own full_sort_predict / full_sort_predict_kg matrices and its masked top-10 (tests/golden/make_golden.py)
    through the product form (forced at the goldens' small sizes), at the north star's 1e-4; ranked columns identical wherever
    the reference's neighbouring scores are further apart than the tolerance."""
    from tests.test_gpu_kge import ATOL, RTOL, golden_case

    z, model, tables, inter = golden_case(path)
    d = next(iter(tables.values())).shape[1]
    nI = z["fs.scores"].shape[1]
    m = build(model, tables, d, 20, nI, 48, 5, gemm_min_scores=0)
    users = torch.from_numpy(z["fs.users"]).to(dev())
    with torch.no_grad():
        assert m._use_gemm_full_sort(users.numel(), nI)
        fs = m.full_sort_predict({"user_id": users})
        np.testing.assert_allclose(fs.cpu().numpy(), z["fs.scores"], rtol=RTOL, atol=ATOL)
        fskg = m.full_sort_predict_kg({"head_id": torch.from_numpy(z["fskg.heads"]).to(dev()),
                                       "relation_id": torch.from_numpy(z["fskg.rels"]).to(dev())})
        np.testing.assert_allclose(fskg.cpu().numpy(), z["fskg.scores"], rtol=RTOL, atol=ATOL)
        hist = (torch.from_numpy(z["eval.history_rows"]).to(dev()), torch.from_numpy(z["eval.history_cols"]).to(dev()))
        tv, ti = m.full_sort_topk({"user_id": users}, 10, hist)
    tv, ti = tv.cpu().numpy(), ti.cpu().numpy()
    ref_v, ref_i = z["eval.topk_values"], z["eval.topk_idx"]
    full_sorted = -np.sort(-z["eval.masked"], axis=1)
    checked = 0
    for r in range(ref_v.shape[0]):
        for c in range(10):
            if not np.isfinite(ref_v[r, c]):
                continue
            gap_prev = np.inf if c == 0 else full_sorted[r, c - 1] - full_sorted[r, c]
            gap_next = full_sorted[r, c] - full_sorted[r, c + 1] if c + 1 < full_sorted.shape[1] else np.inf
            tol = ATOL + RTOL * abs(ref_v[r, c])
            if min(gap_prev, gap_next) > 4 * tol:
                assert ti[r, c] == ref_i[r, c], (r, c)
                checked += 1
            np.testing.assert_allclose(tv[r, c], ref_v[r, c], rtol=RTOL, atol=ATOL)
    assert checked > 0


def test_edges_of_the_product_form():
    """Empty batches, k beyond the re-rank's 64 candidates, k = n, a single candidate column, a history that masks almost a whole
    row (fewer live columns than k): the product form returns what the gather form returns."""
    from hopwise_amd import ops

    model, dim, nU, nE, nI, nR = "RotatE", 32, 40, 600, 500, 5
    tables = random_tables(model, nU, nE, nR, dim, seed=2)
    m = build(model, tables, dim, nU, nI, nE, nR, gemm_min_scores=0)
    users = torch.arange(1, 20, device=dev())
    inter = {"user_id": users}
    B = users.numel()

    def both(k, hist=None):
        with torch.no_grad():
            m.gemm_full_sort = True
            g = m.full_sort_topk(inter, k, hist)
            m.gemm_full_sort = False
            r = m.full_sort_topk(inter, k, hist)
            m.gemm_full_sort = True
        return g, r

    for k in (1, 58, 64, 70, 128):  # 58 -> 64 candidates; 70, 128: beyond the re-rank, the finished scores are ranked
        (gv, gi), (kv, ki) = both(k)
        np.testing.assert_allclose(gv.cpu().numpy(), kv.cpu().numpy(), rtol=2e-5, atol=1e-6)
        assert (gi == ki).float().mean() > 0.98
    # a row with three live columns only: the rest of its top-10 is -inf, as the gather form reports it
    cols = torch.arange(0, nI, device=dev())
    keep = torch.tensor([7, 99, 320], device=dev())
    dead = cols[~torch.isin(cols, keep)]
    hist = (torch.full_like(dead, 4), dead)
    (gv, gi), (kv, ki) = both(10, hist)
    assert torch.equal(gi[4, :3].sort().values, keep) and torch.isinf(gv[4, 3:]).all() and torch.isinf(kv[4, 3:]).all()
    assert torch.equal(gi[:4], ki[:4]) and torch.equal(gi[5:], ki[5:])
    # empty batch
    with torch.no_grad():
        ev, ei = m.full_sort_topk({"user_id": users[:0]}, 10)
        assert ev.shape == (0, 10) and ei.shape == (0, 10)
        assert m.full_sort_predict({"user_id": users[:0]}).shape[0] == 0
    # the entry points on nothing
    head, rel, tail = m._tables("rec")
    q, qn = ops.kge_query_rows(model, head, rel, tail, users[:0], None, m._rec_rel_full_sort(), sqnorm=True)
    assert q.shape == (0, 2 * dim) and qn.shape == (0,)
    t = m._cat_tables(tail)[:nI]
    assert ops.gemm_dist(q, t, qn, m._tail_sqnorm(tail)[:nI]).shape == (0, nI)
    z = torch.zeros((0, 16), device=dev())
    v, i, flag = ops.topk_rerank(z.long(), z, z, 10)
    assert v.shape == (0, 10) and int(flag.item()) == 0
    # one candidate column
    one = build(model, tables, dim, nU, 1, nE, nR, gemm_min_scores=0)
    with torch.no_grad():
        s = one.full_sort_predict(inter)
        one.gemm_full_sort = False
        np.testing.assert_allclose(s.cpu().numpy(), one.full_sort_predict(inter).view_as(s).cpu().numpy(), rtol=2e-5, atol=1e-6)


def test_ranking_product_error_bound_and_the_topk_it_feeds():
    """hw_gemm_rank (three split terms of six): every entry within 1.2e-5 sum |q_k t_k| of the float64 product -- the bound the
    re-rank's proof is given -- and the evaluation top-k that ranks by it returns what the six-term path and the gather
    kernel return."""
    from hopwise_amd import ops

    rng = np.random.default_rng(4)
    for M, N, K in ((130, 1003, 400), (64, 4600, 128), (7, 300, 64)):
        q = (rng.standard_normal((M, K)) * rng.uniform(0.1, 3.0, (M, 1))).astype(np.float32)
        t = (rng.standard_normal((N, K)) * rng.uniform(0.1, 3.0, (N, 1))).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        got = ops.gemm_rank(torch.from_numpy(q).to(dev()), torch.from_numpy(t).to(dev()), torch.from_numpy(b).to(dev()))
        want = q.astype(np.float64) @ t.astype(np.float64).T + b
        bound = ops.RANK_PRODUCT_REL_ERR * (np.abs(q).astype(np.float64) @ np.abs(t).astype(np.float64).T)
        err = np.abs(got.cpu().numpy() - want)
        assert (err <= bound + 1e-6 * np.abs(want) + 1e-6).all(), float((err / (bound + 1e-12)).max())
        assert err.max() > 1e-7 * np.abs(want).max()  # and it IS the cheaper product: not the six-term one
    for model in ("RotatE", "TransE"):
        dim, nU, nI, nE, nR, B, k = 64, 60, 4600, 5000, 6, 70, 10
        tables = random_tables(model, nU, nE, nR, dim, seed=3)
        m3 = build(model, tables, dim, nU, nI, nE, nR, gemm_min_scores=0)
        m6 = build(model, tables, dim, nU, nI, nE, nR, gemm_min_scores=0, gemm_rank_terms=6)
        assert m3.gemm_rank_terms == 3 and m6.gemm_rank_terms == 6
        users = torch.from_numpy(rng.integers(0, nU, B)).to(dev())
        hist = (torch.from_numpy(rng.integers(0, B, 60)).to(dev()), torch.from_numpy(rng.integers(0, nI, 60)).to(dev()))
        with torch.no_grad():
            v3, i3 = m3.full_sort_topk({"user_id": users}, k, hist)
            v6, i6 = m6.full_sort_topk({"user_id": users}, k, hist)
            m6.gemm_full_sort = False
            vk, ik = m6.full_sort_topk({"user_id": users}, k, hist)
        np.testing.assert_array_equal(i3.cpu().numpy(), i6.cpu().numpy())
        np.testing.assert_array_equal(v3.cpu().numpy(), v6.cpu().numpy())
        np.testing.assert_array_equal(i3.cpu().numpy(), ik.cpu().numpy())


def test_mask_bits_tile_maxima_and_the_topk_from_them():
    """hw_gemm_rank with its two by-products -- the history mask applied in the epilogue, the maximum of every 64-column stretch --
    and hw_topk_from_tile_max: the top-k of the masked matrix, bit for bit what mask + hw_topk_rows give on the same matrix (ties,
    -inf rows, ragged widths), and the overflow flag where a row's equal scores outgrow the kernel's lists."""
    from hopwise_amd import ops

    rng = np.random.default_rng(9)
    for M, N, K, k in ((130, 4600, 64, 16), (300, 45539, 32, 16), (33, 9000, 128, 24), (64, 8191, 16, 64), (5, 2048, 8, 10)):
        q = rng.standard_normal((M, K)).astype(np.float32)
        t = rng.standard_normal((N, K)).astype(np.float32)
        t[N // 3] = t[N // 3 + 7]  # equal scores in every row: the tie goes to the smaller column
        b = rng.standard_normal(N).astype(np.float32)
        b[N // 3] = b[N // 3 + 7]
        qd, td, bd = (torch.from_numpy(x).to(dev()) for x in (q, t, b))
        hr = torch.from_numpy(rng.integers(0, M, 5 * M)).to(dev())
        hc = torch.from_numpy(rng.integers(0, N, 5 * M)).to(dev())
        keep0 = rng.choice(N, 400, replace=False)  # row 0: everything masked but 400 columns (in more than k stretches)
        gone = np.setdiff1d(np.arange(N), keep0)
        hr = torch.cat([hr, torch.zeros(len(gone), dtype=torch.int64, device=dev())])
        hc = torch.cat([hc, torch.from_numpy(gone).to(dev())])
        bits = ops.mask_bits((hr, hc), M, N)
        tmax = torch.full((M, (N + 63) // 64), 123.0, device=dev())
        p = ops.gemm_rank(qd, td, bd, mask=bits, tile_max=tmax)
        plain = ops.gemm_rank(qd, td, bd).clone()
        ops.mask_scores_(plain, (hr, hc))
        np.testing.assert_array_equal(p.cpu().numpy(), plain.cpu().numpy())
        pn = p.cpu().numpy()
        want_max = np.stack([pn[:, 64 * j : 64 * j + 64].max(1) for j in range((N + 63) // 64)], 1)
        np.testing.assert_array_equal(tmax.cpu().numpy(), want_max)
        v, i, over = ops.topk_from_tile_max(p, tmax, k)
        rv, ri = ops.topk_rows(p.contiguous(), k)
        assert int(over.item()) == 0
        np.testing.assert_array_equal(i.cpu().numpy(), ri.cpu().numpy())
        np.testing.assert_array_equal(v.cpu().numpy(), rv.cpu().numpy())
    # a row of equal scores (a user whose every item is masked: all -inf): more candidates than the kernel's lists hold -> the
    # flag, harmless outputs
    same = torch.full((3, 8192), -float("inf"), device=dev())
    same[2] = 0.0
    same[1] = torch.from_numpy(rng.standard_normal(8192).astype(np.float32)).to(dev())
    tm = same.view(3, 128, 64).max(2).values.contiguous()
    v, i, over = ops.topk_from_tile_max(same, tm, 16)
    assert int(over.item()) == 1
    rv, ri = ops.topk_rows(same, 16)
    np.testing.assert_array_equal(i[1].cpu().numpy(), ri[1].cpu().numpy())  # the other rows are answered
    assert int(i[0].min()) >= 0 and int(i[0].max()) < 8192


def test_ranking_product_from_plane_images_equals_the_one_from_fp32_operands():
    """hw_gemm_rank_planes (operands split once into bf16 plane images, 256 x 256 tiles fed by DMA) against hw_gemm_rank on the
    matrices the images were made from: scores, masked entries and stretch maxima bit for bit, ragged shapes included."""
    from hopwise_amd import ops

    rng = np.random.default_rng(12)
    for M, N, K in ((300, 1000, 400), (2048, 4600, 200), (17, 257, 64), (256, 256, 16), (1, 70, 20), (513, 777, 130)):
        q = rng.standard_normal((M, K)).astype(np.float32)
        t = rng.standard_normal((N, K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        Kp = (K + 3) & ~3
        qd = torch.zeros((M, Kp), device=dev())[:, :K]
        qd.copy_(torch.from_numpy(q))
        td = torch.zeros((N, Kp), device=dev())[:, :K]
        td.copy_(torch.from_numpy(t))
        bd = torch.from_numpy(b).to(dev())
        hist = (torch.from_numpy(rng.integers(0, M, 3 * M)).to(dev()), torch.from_numpy(rng.integers(0, N, 3 * M)).to(dev()))
        bits = ops.mask_bits(hist, M, N)
        nt = (N + 63) // 64
        tm1, tm2 = torch.full((M, nt), 5.0, device=dev()), torch.full((M, nt), 7.0, device=dev())
        want = ops.gemm_rank(qd, td, bd, mask=bits, tile_max=tm1)
        assert want is not None
        got = ops.gemm_rank_planes(ops.rank_planes(qd), ops.rank_planes(td), M, N, K, bd, mask=bits, tile_max=tm2)
        np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy(), err_msg=f"{M} {N} {K}")
        np.testing.assert_array_equal(tm2.cpu().numpy(), tm1.cpu().numpy())
        plain = ops.gemm_rank_planes(ops.rank_planes(qd), ops.rank_planes(td), M, N, K)
        np.testing.assert_array_equal(plain.cpu().numpy(), ops.gemm_rank(qd, td).cpu().numpy())


def test_evaluation_topk_through_the_plane_image_product_is_the_gather_paths():
    """B >= 512 and K >= 128: the evaluation top-k takes the plane-image product (cached table planes, per-call query planes);
    columns and scores equal the gather kernel's, and follow the weights after an optimizer step."""
    from hopwise_amd.optim import FusedAdam

    rng = np.random.default_rng(21)
    model, dim, nU, nI, nE, nR, B, k = "RotatE", 64, 700, 4600, 5000, 6, 600, 10
    m = build(model, random_tables(model, nU, nE, nR, dim, seed=5), dim, nU, nI, nE, nR, gemm_min_scores=0)
    assert m.gemm_rank_planes and 2 * dim >= m.RANK_PLANES_MIN_K
    users = torch.from_numpy(rng.integers(0, nU, B)).to(dev())
    hist = (torch.from_numpy(rng.integers(0, B, 3000)).to(dev()), torch.from_numpy(rng.integers(0, nI, 3000)).to(dev()))
    for step in range(2):
        with torch.no_grad():
            v, i = m.full_sort_topk({"user_id": users}, k, hist)
            assert any(s[0] == "rankplanes" for s in m._gemm_tables)
            m.gemm_full_sort = False
            vk, ik = m.full_sort_topk({"user_id": users}, k, hist)
            m.gemm_full_sort = True
        np.testing.assert_array_equal(i.cpu().numpy(), ik.cpu().numpy())
        np.testing.assert_array_equal(v.cpu().numpy(), vk.cpu().numpy())
        opt = FusedAdam(m.parameters(), lr=0.05)  # move the tables: the cached planes must follow
        g = torch.Generator().manual_seed(step)
        for p in m.parameters():
            p.grad.copy_(torch.randn(p.shape, generator=g).to(p.device))
        opt.mark_touched()
        opt.step()
