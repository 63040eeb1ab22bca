"""HIP scorers / losses / gradients against (1) vectors produced by the reference and (2) the numpy oracle.

Tolerances: scores 1e-4 relative (north star) with 1e-5 absolute floor; gradients 2e-4 relative with an
absolute floor scaled to the gradient magnitude (atomics reorder float adds); ranked indices exact where
the oracle's adjacent score gap exceeds the score tolerance.
"""

import glob
import os

import numpy as np
import pytest
import torch

from oracle import kge as okge
from tests.helpers import FakeDataset, load_tables, make_config, random_tables

pytestmark = pytest.mark.gpu
OOB_BASE = {}


@pytest.fixture(autouse=True)
def _oob_baseline():
    from hopwise_amd import ops as _ops

    OOB_BASE["v"] = _ops.oob_count(torch.device("cuda", 0))
    yield

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KGE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "kge_[A-Z]*_d*.npz")))
RTOL, ATOL = 1e-4, 1e-5
ALL_MODELS = okge.MODELS + okge.DENSE_MODELS


def dev():
    return torch.device("cuda", 0)


def get_cls(name):
    import hopwise_amd.model as M

    return getattr(M, name)


def golden_case(path):
    z = np.load(path)
    tables = {k[len("table.") : -len(".weight")]: z[k] for k in z.files if k.startswith("table.")}
    inter = {k[len("inter.") :]: z[k] for k in z.files if k.startswith("inter.")}
    model = os.path.basename(path).split("_")[1]
    return z, model, tables, inter


def build(model, tables, d, nU, nI, nE, nR, **extra):
    m = get_cls(model)(make_config(dev(), d, 1.0, **extra), FakeDataset(nU, nI, nE, nR)).to(dev())
    return load_tables(m, tables, dev())


def to_dev(inter):
    return {k: torch.from_numpy(v).to(dev()) for k, v in inter.items()}


@pytest.mark.parametrize("path", KGE_FILES, ids=[os.path.basename(p)[4:-4] for p in KGE_FILES])
def test_against_reference_golden(path):
    z, model, tables, inter = golden_case(path)
    d = next(iter(tables.values())).shape[1]
    nI = z["fs.scores"].shape[1]
    m = build(model, tables, d, 20, nI, 48, 5)
    it = to_dev(inter)
    with torch.no_grad():
        np.testing.assert_allclose(m.predict(it).cpu().numpy(), z["predict"], rtol=RTOL, atol=ATOL)
        if "predict_kg" in z.files:
            np.testing.assert_allclose(m.predict_kg(it).cpu().numpy(), z["predict_kg"], rtol=RTOL, atol=ATOL)
        else:
            with pytest.raises(NotImplementedError):  # TransH: the reference defines no kg-side predictors
                m.predict_kg(it)
    m.zero_grad()
    loss = m.calculate_loss(it)
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["loss"], rtol=1e-5)
    for k, p in m.named_parameters():
        ref = z[f"grad.{k}"]
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * scale + 1e-9, err_msg=k)
    with torch.no_grad():
        fs = m.full_sort_predict({"user_id": torch.from_numpy(z["fs.users"]).to(dev())})
        assert tuple(fs.shape) == z["fs.scores"].shape
        np.testing.assert_allclose(fs.cpu().numpy(), z["fs.scores"], rtol=RTOL, atol=ATOL)
        if "fskg.scores" in z.files:
            fskg = m.full_sort_predict_kg(
                {"head_id": torch.from_numpy(z["fskg.heads"]).to(dev()), "relation_id": torch.from_numpy(z["fskg.rels"]).to(dev())}
            )
            np.testing.assert_allclose(fskg.cpu().numpy(), z["fskg.scores"], rtol=RTOL, atol=ATOL)
    # evaluation protocol on device: mask + top-10; indices must equal the reference's wherever the
    # reference's neighbouring scores are further apart than the score tolerance
    from hopwise_amd import ops

    masked = ops.mask_scores_(fs.clone(), (torch.from_numpy(z["eval.history_rows"]).to(dev()),
                                           torch.from_numpy(z["eval.history_cols"]).to(dev())))
    assert np.array_equal(np.isneginf(masked.cpu().numpy()), np.isneginf(z["eval.masked"]))
    tv, ti = ops.topk_rows(masked, 10)
    tv, ti = tv.cpu().numpy(), ti.cpu().numpy()
    ref_v, ref_i = z["eval.topk_values"], z["eval.topk_idx"]
    full_sorted = -np.sort(-z["eval.masked"], axis=1)
    for r in range(ref_v.shape[0]):
        for c in range(10):
            if not np.isfinite(ref_v[r, c]):
                continue
            gap_prev = np.inf if c == 0 else full_sorted[r, c - 1] - full_sorted[r, c]
            gap_next = full_sorted[r, c] - full_sorted[r, c + 1] if c + 1 < full_sorted.shape[1] else np.inf
            tol = ATOL + RTOL * abs(ref_v[r, c])
            if min(gap_prev, gap_next) > 4 * tol:
                assert ti[r, c] == ref_i[r, c], (r, c)
            np.testing.assert_allclose(tv[r, c], ref_v[r, c], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("model", okge.MODELS)
@pytest.mark.parametrize("dim", [4, 7, 50, 100, 130, 200, 256, 500, 1024])
def test_grouped_scorer_vs_oracle(model, dim):
    nU, nE, nR, B, K = 300, 2000, 9, 37, 19
    tables = random_tables(model, nU, nE, nR, dim, seed=dim)
    rng = np.random.default_rng(dim + 1)
    m = build(model, tables, dim, nU, 100, nE, nR)
    from hopwise_amd import ops

    for side, n_head in (("rec", nU), ("kg", nE)):
        h = rng.integers(0, n_head, B)
        r = rng.integers(0, nR, B)
        t = rng.integers(0, nE, (B, 1 + K))
        ref = okge.score_grouped(model, tables, side, h, r, t, margin=m.margin)
        head, rel, tail = m._tables(side)
        got = ops.kge_score_grouped(model, head, rel, tail, torch.from_numpy(h).to(dev()), torch.from_numpy(r).to(dev()),
                                    0, torch.from_numpy(t).to(dev()), m.margin)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
        # j-major output layout + strided (transposed) index input give the same numbers
        tt = torch.from_numpy(np.ascontiguousarray(t.T)).to(dev()).t()
        got2 = ops.kge_score_grouped(model, head, rel, tail, torch.from_numpy(h).to(dev()),
                                     torch.from_numpy(r).to(dev()), 0, tt, m.margin, 0.0, True)
        assert got2.stride() == (1, B)
        np.testing.assert_array_equal(got2.detach().cpu().numpy(), got.detach().cpu().numpy())
    assert ops.oob_count(dev()) == OOB_BASE.get("v", 0)  # cumulative per device: compare with the value at import


@pytest.mark.parametrize("model", ALL_MODELS)
def test_full_sort_vs_oracle_ragged(model):
    # sizes chosen so that chunks, head blocks and sub-wave groups all have ragged tails
    for dim, nE, B in ((64, 1003, 1), (200, 777, 5), (100, 130, 9), (24, 67, 2)):
        nU, nR, nI = 50, 6, min(nE, 61)
        tables = random_tables(model, nU, nE, nR, dim, seed=dim + B)
        m = build(model, tables, dim, nU, nI, nE, nR)
        rng = np.random.default_rng(3)
        users = rng.integers(0, nU, B)
        heads = rng.integers(0, nE, B)
        rels = rng.integers(0, nR, B)
        with torch.no_grad():
            fs = m.full_sort_predict({"user_id": torch.from_numpy(users).to(dev())}).cpu().numpy()
            fk = m.full_sort_predict_kg({"head_id": torch.from_numpy(heads).to(dev()),
                                         "relation_id": torch.from_numpy(rels).to(dev())}).cpu().numpy()
        ui = nR - 1
        np.testing.assert_allclose(fs, okge.full_sort_predict(model, tables, users, nI, m.margin, ui), rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(fk, okge.full_sort_predict_kg(model, tables, heads, rels, m.margin), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("model", okge.MODELS)
@pytest.mark.parametrize("dim", [16, 200, 300])
def test_loss_and_grads_vs_oracle(model, dim):
    nU, nI, nE, nR, Br, Bk = 40, 30, 90, 7, 64, 96
    tables = random_tables(model, nU, nE, nR, dim, seed=11)
    rng = np.random.default_rng(5)
    inter = dict(
        user_id=rng.integers(0, nU, Br), item_id=rng.integers(0, nI, Br), neg_item_id=rng.integers(0, nI, Br),
        head_id=rng.integers(0, nE, Bk), relation_id=rng.integers(0, nR, Bk), tail_id=rng.integers(0, nE, Bk),
        neg_tail_id=rng.integers(0, nE, Bk),
    )
    m = build(model, tables, dim, nU, nI, nE, nR)
    loss_ref, grads_ref = okge.calculate_loss(model, tables, inter, m.margin if model != "ComplEx" else 1.0,
                                              ui_relation=nR - 1, with_grad=True)
    m.zero_grad()
    loss = m.calculate_loss(to_dev(inter))
    loss.backward()
    np.testing.assert_allclose(loss.item(), loss_ref, rtol=1e-5)
    for k, p in m.named_parameters():
        ref = grads_ref[k[: -len(".weight")]]
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * scale + 1e-9, err_msg=k)


@pytest.mark.parametrize("model", ALL_MODELS + okge.EXT_MODELS)
def test_three_adam_steps_match_reference(model):
    """A18: zero_grad -> calculate_loss -> backward -> Adam.step, compared with the reference run."""
    z = np.load(os.path.join(GOLDEN, f"kge_train_{model}.npz"))
    tables = {k[len("init.") : -len(".weight")]: z[k] for k in z.files if k.startswith("init.")}
    m = build(model, tables, 16, 20, 16, 48, 5)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0.0)
    m.train()
    losses = []
    for step in range(3):
        inter = {k[len(f"step{step}.") :]: z[k] for k in z.files if k.startswith(f"step{step}.")}
        opt.zero_grad()
        loss = m.calculate_loss(to_dev(inter))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, z["losses"], rtol=2e-5)
    for k, v in m.state_dict().items():
        # Adam normalises the step, so a 1e-7 gradient difference can move a weight by up to lr; the
        # bound is a fraction of the 3 * lr a weight can move at all; most weights must agree to 2e-5 (80%: the
        # relation tables here hold only 80 weights, a handful of near-zero gradients moves the fraction a lot).
        np.testing.assert_allclose(v.cpu().numpy(), z[f"final.{k}"], rtol=0, atol=2e-4, err_msg=k)
        assert (np.abs(v.cpu().numpy() - z[f"final.{k}"]) < 2e-5).mean() > 0.8, k


def test_losses_vs_oracle():
    from hopwise_amd import ops

    rng = np.random.default_rng(0)
    for n in (1, 7, 300, 70001):
        x = (rng.standard_normal(n) * 4).astype(np.float32)
        npos = n // 3
        y = (np.arange(n) < npos).astype(np.float32)
        xt = torch.from_numpy(x).to(dev()).requires_grad_(True)
        l = ops.bce_logits_mean(xt, npos)
        l.backward()
        np.testing.assert_allclose(l.item(), okge.bce_with_logits_mean(x, y), rtol=2e-6)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), okge.bce_with_logits_grad(x, y), rtol=1e-5, atol=1e-9)
        p = rng.standard_normal(n).astype(np.float32)
        q = rng.standard_normal(n).astype(np.float32)
        pt = torch.from_numpy(p).to(dev()).requires_grad_(True)
        qt = torch.from_numpy(q).to(dev()).requires_grad_(True)
        l = ops.margin_ranking_mean(pt, qt, 0.7)
        l.backward()
        np.testing.assert_allclose(l.item(), okge.margin_ranking_mean(p, q, 0.7), rtol=2e-6)
        act = ((-(p.astype(np.float64) - q) + 0.7) > 0) / n
        np.testing.assert_allclose(pt.grad.cpu().numpy(), -act, rtol=1e-6)
        np.testing.assert_allclose(qt.grad.cpu().numpy(), act, rtol=1e-6)
    for B, K in ((1, 1), (5, 3), (300, 256), (2048, 70)):
        s = (rng.standard_normal((B, 1 + K)) * 3).astype(np.float32)
        st = torch.from_numpy(s).to(dev()).requires_grad_(True)
        l = ops.self_adversarial_mean(st, 0.5)
        l.backward()
        np.testing.assert_allclose(l.item(), okge.self_adversarial_loss(s, 0.5), rtol=1e-5)
        np.testing.assert_allclose(st.grad.cpu().numpy(), okge.self_adversarial_grad(s, 0.5), rtol=1e-4, atol=1e-9)


def test_self_adversarial_known_answer():
    """Closed form (A9 is unpinned by the reference): with equal negative scores the weights are uniform."""
    from hopwise_amd import ops

    s = torch.tensor([[0.0, 1.0, 1.0, 1.0, 1.0]], device=dev())
    want = np.log(2.0) + np.log1p(np.e)  # -logsig(0) + 4 * (1/4) * -logsig(-1)
    assert abs(ops.self_adversarial_mean(s, 2.0).item() - want) < 1e-6


def test_topk_ties_and_short_rows():
    from hopwise_amd import ops

    s = torch.tensor([[1.0, 3.0, 3.0, -np.inf, 2.0], [0.0, 0.0, 0.0, 0.0, 0.0]], device=dev())
    v, i = ops.topk_rows(s, 4)
    assert i.cpu().tolist() == [[1, 2, 4, 0], [0, 1, 2, 3]]
    v, i = ops.topk_rows(s[:, :3], 5)  # k > columns, strided rows
    assert i.cpu().tolist()[0] == [1, 2, 0, -1, -1]
    rng = np.random.default_rng(1)
    big = rng.standard_normal((3, 13805)).astype(np.float32)
    v, i = ops.topk_rows(torch.from_numpy(big).to(dev()), 10)
    rv, ri = okge.topk_desc(big, 10)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)


def test_mask_scores_with_and_without_the_column0_rule():
    """hw_mask_scores_ex: a shard that does not own entity 0 masks its history entries only; entries outside the matrix (the
    sharded wrapper writes column -1 for ids owned elsewhere) are ignored."""
    from hopwise_amd import ops

    rng = np.random.default_rng(3)
    a = rng.standard_normal((6, 50)).astype(np.float32)
    hr = np.array([0, 0, 5, 2, 7, 3, 1], np.int64)
    hc = np.array([0, 7, 49, -1, 3, 50, 0], np.int64)
    hist = (torch.from_numpy(hr).to(dev()), torch.from_numpy(hc).to(dev()))
    for column0 in (True, False):
        want = a.copy()
        if column0:
            want[:, 0] = -np.inf
        for r, c in zip(hr, hc):
            if 0 <= r < 6 and 0 <= c < 50:
                want[r, c] = -np.inf
        got = ops.mask_scores_(torch.from_numpy(a).to(dev()), hist, column0=column0)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    got = ops.mask_scores_(torch.from_numpy(a).to(dev()), None, column0=False)  # nothing to do: no launch
    np.testing.assert_array_equal(got.cpu().numpy(), a)


def test_cpu_tensors_are_rejected():
    from hopwise_amd import _lib, ops

    with pytest.raises(_lib.HopwiseAmdError):
        ops.topk_rows(torch.zeros(2, 3), 1)


def test_self_adversarial_model_path():
    """RotatE with self_adversarial=True on a tiled (P positives x K negatives) batch vs the oracle."""
    model, dim, nU, nI, nE, nR, P, K, Bk = "RotatE", 32, 30, 25, 60, 5, 6, 8, 16
    tables = random_tables(model, nU, nE, nR, dim, seed=2)
    rng = np.random.default_rng(9)
    users, pos = rng.integers(1, nU, P), rng.integers(1, nI, P)
    neg = rng.integers(1, nI, (K, P))
    inter = dict(user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=neg.reshape(-1),
                 head_id=rng.integers(1, nE, Bk), relation_id=rng.integers(1, nR, Bk), tail_id=rng.integers(1, nE, Bk),
                 neg_tail_id=rng.integers(1, nE, Bk))
    m = build(model, tables, dim, nU, nI, nE, nR, self_adversarial=True, adversarial_temperature=0.7,
              train_neg_sample_args={"sample_num": K})
    loss = m.calculate_loss(to_dev(inter))
    tails = np.concatenate([pos[:, None], neg.T], axis=1)
    s = okge.score_grouped(model, tables, "rec", users, np.full(P, nR - 1), tails, margin=1.0)
    kgp = okge.score_rows(model, (tables["entity_embedding"][inter["head_id"]], tables["entity_embedding_im"][inter["head_id"]]),
                          (tables["relation_embedding"][inter["relation_id"]],),
                          (tables["entity_embedding"][inter["tail_id"]], tables["entity_embedding_im"][inter["tail_id"]]), 1.0)
    kgn = okge.score_rows(model, (tables["entity_embedding"][inter["head_id"]], tables["entity_embedding_im"][inter["head_id"]]),
                          (tables["relation_embedding"][inter["relation_id"]],),
                          (tables["entity_embedding"][inter["neg_tail_id"]], tables["entity_embedding_im"][inter["neg_tail_id"]]), 1.0)
    want = okge.self_adversarial_loss(s, 0.7) + okge.bce_with_logits_mean(
        np.concatenate([kgp, kgn]), np.concatenate([np.ones(Bk), np.zeros(Bk)]))
    np.testing.assert_allclose(loss.item(), want, rtol=1e-5)
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_row_primitives():
    from hopwise_amd import ops

    rng = np.random.default_rng(0)
    for dim in (3, 64, 200):
        tab = rng.standard_normal((97, dim)).astype(np.float32)
        idx = rng.integers(0, 97, 1000)
        t = torch.from_numpy(tab).to(dev())
        out = ops.gather_rows(t, torch.from_numpy(idx).to(dev()))
        np.testing.assert_array_equal(out.cpu().numpy(), tab[idx])
        src = rng.standard_normal((1000, dim)).astype(np.float32)
        acc = torch.zeros_like(t)
        ops.scatter_add_rows_(acc, torch.from_numpy(idx).to(dev()), torch.from_numpy(src).to(dev()))
        want = np.zeros_like(tab, dtype=np.float64)
        np.add.at(want, idx, src)
        np.testing.assert_allclose(acc.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("model", ["RotatE", "ComplEx", "TransE"])
def test_row_sharded_wrapper_single_rank_matches_plain_model(model):
    """The compact-table path (lookup -> remap -> same kernels) on one rank equals the plain model."""
    from hopwise_amd import parallel as par

    dim, nU, nI, nE, nR, Br, Bk = 64, 40, 30, 90, 7, 48, 80
    tables = random_tables(model, nU, nE, nR, dim, seed=4)
    rng = np.random.default_rng(6)
    inter = dict(
        user_id=rng.integers(0, nU, Br), item_id=rng.integers(0, nI, Br), neg_item_id=rng.integers(0, nI, Br),
        head_id=rng.integers(0, nE, Bk), relation_id=rng.integers(0, nR, Bk), tail_id=rng.integers(0, nE, Bk),
        neg_tail_id=rng.integers(0, nE, Bk),
    )
    plain = build(model, tables, dim, nU, nI, nE, nR)
    loss_p = plain.calculate_loss(to_dev(inter))
    loss_p.backward()
    sh = par.RowShardedKGE(build(model, tables, dim, nU, nI, nE, nR))
    loss_s = sh.calculate_loss(to_dev(inter))
    loss_s.backward()
    np.testing.assert_allclose(loss_s.item(), loss_p.item(), rtol=1e-6)
    for n in plain.USER_TABLES + plain.ENTITY_TABLES:
        a = getattr(sh, "shard_" + n).grad.cpu().numpy()
        b = getattr(plain, n).weight.grad.cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6 * (np.abs(b).max() + 1e-12) + 1e-9, err_msg=n)
    for n in plain.RELATION_TABLES:
        np.testing.assert_allclose(getattr(sh.model, n).weight.grad.cpu().numpy(),
                                   getattr(plain, n).weight.grad.cpu().numpy(), rtol=2e-4, atol=1e-8)


@pytest.mark.parametrize("model", okge.MODELS)
@pytest.mark.parametrize("variant", ["0", "1"])
def test_grouped_many_tails_both_kernels(model, variant, monkeypatch):
    """K = 300 tails per group through the plain and the XCD-partitioned kernel (ragged partitions, list carry-over)."""
    from hopwise_amd import ops

    monkeypatch.setenv("HW_GROUPED_VARIANT", variant)
    for dim, B, K in ((200, 9, 300), (64, 5, 131), (100, 3, 700)):
        nU, nE, nR = 50, 3000, 6
        tables = random_tables(model, nU, nE, nR, dim, seed=K)
        m = build(model, tables, dim, nU, 100, nE, nR)
        rng = np.random.default_rng(K)
        h, r = rng.integers(0, nE, B), rng.integers(0, nR, B)
        t = rng.integers(0, nE, (B, 1 + K))
        t[0, :40] = 8 * rng.integers(0, 300, 40) + 3  # one partition over-full, others starved
        head, rel, tail = m._tables("kg")
        got = ops.kge_score_grouped(model, head, rel, tail, torch.from_numpy(h).to(dev()), torch.from_numpy(r).to(dev()),
                                    0, torch.from_numpy(t).to(dev()), m.margin)
        ref = okge.score_grouped(model, tables, "kg", h, r, t, margin=m.margin)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("dim", [4, 7, 50, 64, 100])
def test_rescal_grouped_loss_and_grads_vs_oracle(dim):
    """RESCAL (h^T M_r t): grouped scores, MarginRankingLoss and all gradients (incl. the d x d relation matrices)."""
    from hopwise_amd import ops

    model, nU, nI, nE, nR, Br, Bk = "RESCAL", 40, 30, 90, 5, 24, 40
    tables = random_tables(model, nU, nE, nR, dim, seed=dim)
    m = build(model, tables, dim, nU, nI, nE, nR)
    rng = np.random.default_rng(dim)
    h, r, t = rng.integers(0, nE, 11), rng.integers(0, nR, 11), rng.integers(0, nE, (11, 9))
    head, rel, tail = m._tables("kg")
    got = ops.kge_score_grouped(model, head, rel, tail, torch.from_numpy(h).to(dev()), torch.from_numpy(r).to(dev()), 0,
                                torch.from_numpy(t).to(dev()), 0.0)
    ref = okge.score_grouped(model, tables, "kg", h, r, t)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    inter = dict(
        user_id=rng.integers(0, nU, Br), item_id=rng.integers(0, nI, Br), neg_item_id=rng.integers(0, nI, Br),
        head_id=rng.integers(0, nE, Bk), relation_id=rng.integers(0, nR, Bk), tail_id=rng.integers(0, nE, Bk),
        neg_tail_id=rng.integers(0, nE, Bk),
    )
    loss_ref, grads_ref = okge.calculate_loss(model, tables, inter, 1.0, ui_relation=nR - 1, with_grad=True)
    m.zero_grad()
    loss = m.calculate_loss(to_dev(inter))
    loss.backward()
    np.testing.assert_allclose(loss.item(), loss_ref, rtol=1e-5)
    for k, p in m.named_parameters():
        ref = grads_ref[k[: -len(".weight")]]
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * scale + 1e-9, err_msg=k)


@pytest.mark.parametrize("model", ["RotatE", "ComplEx"])
def test_tiled_rec_batch_scored_as_groups_equals_row_by_row(model):
    """sample_num = K > 1: the dataloader tiles each positive K times (abstract_dataloader.py:192-198).  The BCE models
    score each positive once per group with weight K; loss and every gradient must equal the row-by-row evaluation of
    the same batch (what the reference computes)."""
    import copy

    import hopwise_amd.model as M
    from tests.helpers import FakeDataset, make_config

    nU, nI, nE, nR, d, K, P = 50, 30, 80, 5, 20, 6, 7
    torch.manual_seed(11)
    cfg = make_config(dev(), d, 3.0, train_neg_sample_args={"sample_num": K})
    m1 = getattr(M, model)(cfg, FakeDataset(nU, nI, nE, nR)).to(dev())
    m2 = copy.deepcopy(m1)
    m2.rec_tiled_layout = False
    rng = np.random.default_rng(3)
    users, pos = rng.integers(1, nU, P), rng.integers(1, nI, P)
    B = P * K
    inter = to_dev(dict(
        user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B),
        head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
        neg_tail_id=rng.integers(1, nE, B)))
    l1, l2 = m1.calculate_loss(inter), m2.calculate_loss(inter)
    l1.backward()
    l2.backward()
    np.testing.assert_allclose(l1.item(), l2.item(), rtol=2e-6)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        scale = p2.grad.abs().max().item()
        np.testing.assert_allclose(p1.grad.cpu().numpy(), p2.grad.cpu().numpy(), rtol=2e-4, atol=2e-6 * scale, err_msg=k)


@pytest.mark.parametrize("model", ["TransE", "DistMult", "RESCAL", "RotatE", "ComplEx", "TorusE", "TransH", "TransD",
                                   "Analogy", "HolE"])
@pytest.mark.parametrize("K", [1, 4])
def test_fused_loss_node_equals_the_per_op_graph(model, K):
    """calculate_loss as one autograd node (ops.FusedPairLoss, the default) against the per-op graph it replaces (scorer
    nodes, stack / cat, loss nodes): same loss, same gradients, for every pairwise gather model, pair and tiled batches."""
    import copy

    import hopwise_amd.model as M
    from tests.helpers import FakeDataset, make_config

    if K > 1 and model not in ("RotatE", "ComplEx"):
        pytest.skip("only the BCE models regroup tiled batches")
    nU, nI, nE, nR, d, P = 50, 30, 80, 5, 16, 9
    torch.manual_seed(21)
    cfg = make_config(dev(), d, 2.0, train_neg_sample_args={"sample_num": K})
    m1 = getattr(M, model)(cfg, FakeDataset(nU, nI, nE, nR)).to(dev())
    m2 = copy.deepcopy(m1)
    m2.fused_loss = False
    assert m1.fused_loss
    rng = np.random.default_rng(4)
    B = P * K
    users, pos = rng.integers(1, nU, P), rng.integers(1, nI, P)
    inter = to_dev(dict(
        user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B),
        head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
        neg_tail_id=rng.integers(1, nE, B)))
    l1, l2 = m1.calculate_loss(inter), m2.calculate_loss(inter)
    (3.0 * l1).backward()  # an upstream factor other than 1
    (3.0 * l2).backward()
    np.testing.assert_allclose(l1.item(), l2.item(), rtol=2e-6)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p2.grad is None:
            assert p1.grad is None or float(p1.grad.abs().max()) == 0.0, k
            continue
        scale = p2.grad.abs().max().item()
        np.testing.assert_allclose(p1.grad.cpu().numpy(), p2.grad.cpu().numpy(), rtol=2e-4, atol=2e-6 * scale + 1e-12,
                                   err_msg=k)


@pytest.mark.parametrize("fixed_rel", [False, True])
@pytest.mark.parametrize("dim", [24, 72])
def test_rescal_grouped_grads_with_split_tasks(dim, fixed_rel):
    """Few groups with many tails: the backward splits every group over several wave tasks, whose dL/dQ rows feed the
    head-gradient and relation-matrix kernels (rel_matrix.h); checked against float64 autograd of h^T M_r t."""
    from hopwise_amd import ops

    nE, nR, B, J = 50, 4, 6, 70
    g = torch.Generator().manual_seed(dim)
    E = torch.randn(nE, dim, generator=g)
    M = torch.randn(nR, dim * dim, generator=g) / dim
    h, t = torch.randint(0, nE, (B,), generator=g), torch.randint(0, nE, (B, J), generator=g)
    r = torch.randint(0, nR, (B,), generator=g)
    w = torch.randn(B, J, generator=g)
    w[1] = 0.0  # a group whose tails all carry zero gradient
    E64, M64 = E.double().requires_grad_(), M.double().requires_grad_()
    rr = torch.full((B,), nR - 1) if fixed_rel else r
    q = torch.einsum("bi,bij->bj", E64[h], M64[rr].view(B, dim, dim))
    ref = torch.einsum("bj,bkj->bk", q, E64[t])
    (ref * w.double()).sum().backward()
    Ed, Md = E.to(dev()).requires_grad_(), M.to(dev()).requires_grad_()
    got = ops.kge_score_grouped("RESCAL", [Ed], [Md], [Ed], h.to(dev()), None if fixed_rel else r.to(dev()), nR - 1,
                                t.to(dev()), 0.0)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    (got * w.to(dev())).sum().backward()
    for name, a, b in (("entity", Ed.grad, E64.grad), ("relation", Md.grad, M64.grad)):
        scale = b.abs().max().item()
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=2e-4, atol=2e-6 * scale, err_msg=name)


@pytest.mark.parametrize("model", ALL_MODELS)
def test_fused_full_sort_topk_equals_unfused(model):
    """hw_kge_full_sort_topk == full sort -> mask -> top-k, values and indices, rec and kg side."""
    from hopwise_amd import ops

    for dim, nE, nI, B, k in ((64, 1003, 700, 1, 10), (200 if model != "RESCAL" else 48, 777, 300, 5, 100), (24, 200, 130, 9, 128)):
        nU, nR = 50, 6
        tables = random_tables(model, nU, nE, nR, dim, seed=dim + B)
        m = build(model, tables, dim, nU, nI, nE, nR)
        m.gemm_full_sort = False  # this test is about hw_kge_full_sort_topk (the GEMM form: test_gpu_fullsort_gemm.py)
        rng = np.random.default_rng(B)
        users = torch.from_numpy(rng.integers(0, nU, B)).to(dev())
        heads = torch.from_numpy(rng.integers(0, nE, B)).to(dev())
        rels = torch.from_numpy(rng.integers(0, nR, B)).to(dev())
        hist = (torch.from_numpy(rng.integers(0, B, 40)).to(dev()), torch.from_numpy(rng.integers(0, nI, 40)).to(dev()))
        with torch.no_grad():
            for kg, inter, n in ((False, {"user_id": users}, nI), (True, {"head_id": heads, "relation_id": rels}, nE)):
                full = (m.full_sort_predict_kg(inter) if kg else m.full_sort_predict(inter)).view(B, n).contiguous()
                ops.mask_scores_(full, hist)
                rv, ri = ops.topk_rows(full, min(k, n))
                fv, fi = m.full_sort_topk(inter, min(k, n), hist, kg=kg)
                np.testing.assert_array_equal(fi.cpu().numpy(), ri.cpu().numpy())
                np.testing.assert_array_equal(fv.cpu().numpy(), rv.cpu().numpy())
