"""The widened scorer set (TorusE, TransH, TransD, Analogy, HolE) against the oracle at shapes the
goldens do not cover: every row-shape variant of the kernels, K negatives per group, ragged full sorts, gradients of
all tables, the fused masked top-k.  The reference goldens themselves run through tests/test_gpu_kge.py."""

import numpy as np
import pytest
import torch

from oracle import kge as okge
from tests.helpers import random_tables
from tests.test_gpu_kge import ATOL, RTOL, build, dev, to_dev

pytestmark = pytest.mark.gpu
OOB_BASE = {}


@pytest.fixture(autouse=True)
def _oob_baseline():
    from hopwise_amd import ops as _ops

    OOB_BASE["v"] = _ops.oob_count(torch.device("cuda", 0))
    yield
EXT = tuple(m for m in okge.EXT_MODELS if m != "TransR")  # members of the gather-kernel template
NO_KG_FULL_SORT = ("TransH", "TransR")


def tol_for(model):
    # TorusE sums min(x^2, 1 - x^2) of O(1) terms: absolute error grows with d, relative stays ~1e-6
    return dict(rtol=RTOL, atol=ATOL * (20 if model == "TorusE" else 1))


@pytest.mark.parametrize("model", EXT)
@pytest.mark.parametrize("dim", [4, 7, 50, 100, 130, 200, 256, 500])
def test_grouped_scorer_vs_oracle(model, dim):
    from hopwise_amd import ops

    nU, nE, nR, B, K = 120, 900, 9, 23, 11
    tables = random_tables(model, nU, nE, nR, dim, seed=dim)
    rng = np.random.default_rng(dim + 1)
    m = build(model, tables, dim, nU, 100, nE, nR)
    for side, n_head in (("rec", nU), ("kg", nE)):
        h, r = rng.integers(0, n_head, B), rng.integers(0, nR, B)
        t = rng.integers(0, nE, (B, 1 + K))
        ref = okge.score_grouped(model, tables, side, h, r, t, margin=m.margin)
        head, rel, tail = m._tables(side)
        got = ops.kge_score_grouped(model, head, rel, tail, torch.from_numpy(h).to(dev()), torch.from_numpy(r).to(dev()),
                                    0, torch.from_numpy(t).to(dev()), m.margin)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, **tol_for(model))
    assert ops.oob_count(dev()) == OOB_BASE.get("v", 0)  # cumulative per device: compare with the value at import


@pytest.mark.parametrize("model", okge.EXT_MODELS)
def test_full_sort_vs_oracle_ragged(model):
    for dim, nE, B in ((64, 1003, 1), (200 if model != "TransR" else 72, 777, 5), (100, 130, 9), (24, 67, 2)):
        nU, nR, nI = 50, 6, min(nE, 61)
        tables = random_tables(model, nU, nE, nR, dim, seed=dim + B)
        m = build(model, tables, dim, nU, nI, nE, nR)
        rng = np.random.default_rng(3)
        users, heads, rels = rng.integers(0, nU, B), rng.integers(0, nE, B), rng.integers(0, nR, B)
        with torch.no_grad():
            fs = m.full_sort_predict({"user_id": torch.from_numpy(users).to(dev())}).cpu().numpy()
        np.testing.assert_allclose(fs, okge.full_sort_predict(model, tables, users, nI, m.margin, nR - 1), **tol_for(model))
        kg_inter = {"head_id": torch.from_numpy(heads).to(dev()), "relation_id": torch.from_numpy(rels).to(dev())}
        if model in NO_KG_FULL_SORT:
            with pytest.raises(NotImplementedError):
                m.full_sort_predict_kg(kg_inter)
            continue
        with torch.no_grad():
            fk = m.full_sort_predict_kg(kg_inter).cpu().numpy()
        np.testing.assert_allclose(fk, okge.full_sort_predict_kg(model, tables, heads, rels, m.margin), **tol_for(model))


@pytest.mark.parametrize("model,dim", [(m, d) for m in EXT for d in (16, 100, 200, 300)]
                         + [("TransR", d) for d in (5, 16, 64, 100, 130)])
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
    loss_ref, grads_ref = okge.calculate_loss(model, tables, inter, 1.0, ui_relation=nR - 1, with_grad=True)
    m.zero_grad()
    loss = m.calculate_loss(to_dev(inter))
    loss.backward()
    np.testing.assert_allclose(loss.item(), loss_ref, rtol=1e-5)
    for k, p in m.named_parameters():
        ref = grads_ref[k[: -len(".weight")]]
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * scale + 1e-9, err_msg=k)


@pytest.mark.parametrize("model", [x for x in EXT if x != "TorusE"])
def test_grouped_gradients_with_many_negatives(model):
    """K = 7 negatives per group with arbitrary upstream gradients: head / relation sums over the group and tail
    scatter, rec (fixed relation -> replica path) and kg side."""
    from hopwise_amd import ops

    dim, nU, nE, nR, B, K = 72, 30, 200, 5, 300, 7
    tables = random_tables(model, nU, nE, nR, dim, seed=2)
    m = build(model, tables, dim, nU, 50, nE, nR)
    rng = np.random.default_rng(9)
    for side, n_head in (("rec", nU), ("kg", nE)):
        h, r = rng.integers(0, n_head, B), rng.integers(0, nR, B)
        if side == "rec":
            r[:] = nR - 1
        t = rng.integers(0, nE, (B, 1 + K))
        g = rng.standard_normal((B, 1 + K)).astype(np.float32)
        g[rng.random((B, 1 + K)) < 0.2] = 0
        hn, rn, tn = okge.slots(model, side)
        H = tuple(tables[n][h][:, None, :] for n in hn)
        R = tuple(tables[n][r][:, None, :] for n in rn)
        T = tuple(tables[n][t] for n in tn)
        gH, gR, gT = okge.score_rows_vjp(model, H, R, T, g, m.margin, 0.0)
        want = {n: np.zeros(tables[n].shape, np.float64) for n in set(hn + rn + tn)}
        for names, idx, rows in ((hn, h, gH), (rn, r, gR)):
            for nme, gr in zip(names, rows):
                np.add.at(want[nme], idx, np.broadcast_to(gr, (B, 1 + K, dim)).sum(1))
        for nme, gr in zip(tn, gT):
            np.add.at(want[nme], t.reshape(-1), np.broadcast_to(gr, (B, 1 + K, dim)).reshape(-1, dim))
        m.zero_grad()
        head, rel, tail = m._tables(side)
        out = ops.kge_score_grouped(model, head, rel, tail, torch.from_numpy(h).to(dev()),
                                    None if side == "rec" else torch.from_numpy(r).to(dev()), nR - 1,
                                    torch.from_numpy(t).to(dev()), m.margin)
        out.backward(torch.from_numpy(g).to(dev()))
        for nme, ref in want.items():
            got = getattr(m, nme).weight.grad.cpu().numpy()
            scale = np.abs(ref).max() + 1e-12
            np.testing.assert_allclose(got, ref, rtol=3e-4, atol=3e-6 * scale + 1e-9, err_msg=f"{side}:{nme}")


@pytest.mark.parametrize("model", okge.EXT_MODELS)
def test_fused_full_sort_topk_equals_unfused(model):
    from hopwise_amd import ops

    for dim, nE, nI, B, k in ((64, 1003, 700, 1, 10), (200 if model != "TransR" else 48, 777, 300, 5, 100),
                              (24, 200, 130, 9, 128)):
        nU, nR = 50, 6
        tables = random_tables(model, nU, nE, nR, dim, seed=dim + B)
        m = build(model, tables, dim, nU, nI, nE, nR)
        if hasattr(m, "gemm_full_sort"):
            m.gemm_full_sort = False  # this test is about hw_kge_full_sort_topk (the GEMM form: test_gpu_fullsort_gemm.py)
        rng = np.random.default_rng(B)
        users = torch.from_numpy(rng.integers(0, nU, B)).to(dev())
        heads = torch.from_numpy(rng.integers(0, nE, B)).to(dev())
        rels = torch.from_numpy(rng.integers(0, nR, B)).to(dev())
        hist = (torch.from_numpy(rng.integers(0, B, 40)).to(dev()), torch.from_numpy(rng.integers(0, nI, 40)).to(dev()))
        sides = [(False, {"user_id": users}, nI)]
        if model not in NO_KG_FULL_SORT:
            sides.append((True, {"head_id": heads, "relation_id": rels}, nE))
        with torch.no_grad():
            for kg, inter, n in sides:
                full = (m.full_sort_predict_kg(inter) if kg else m.full_sort_predict(inter)).view(B, n).contiguous()
                ops.mask_scores_(full, hist)
                rv, ri = ops.topk_rows(full, min(k, n))
                fv, fi = m.full_sort_topk(inter, min(k, n), hist, kg=kg)
                np.testing.assert_array_equal(fi.cpu().numpy(), ri.cpu().numpy())
                np.testing.assert_array_equal(fv.cpu().numpy(), rv.cpu().numpy())


def test_toruse_scorer_has_no_gradient_path():
    from hopwise_amd import _lib, ops

    tables = random_tables("TorusE", 10, 20, 4, 8, seed=0)
    m = build("TorusE", tables, 8, 10, 10, 20, 4)
    head, rel, tail = m._tables("kg")
    idx = torch.zeros(3, dtype=torch.int64, device=dev())
    out = ops.kge_score_grouped("TorusE", head, rel, tail, idx, idx, 0, idx.view(3, 1), 0.0)
    with pytest.raises(_lib.HopwiseAmdError):
        out.sum().backward()


def test_transr_projection_primitives():
    """hw_kge_project_rows / _grad against numpy: fixed and per-row relations, repeated rows, every dim bucket."""
    from hopwise_amd import ops

    rng = np.random.default_rng(0)
    for d, n_rel, n_ent, n in ((5, 3, 17, 40), (64, 4, 50, 700), (100, 6, 30, 333), (256, 2, 9, 64)):
        mats = torch.from_numpy((rng.standard_normal((n_rel, d * d)) / np.sqrt(d)).astype(np.float32)).to(dev())
        tab = torch.from_numpy(rng.standard_normal((n_ent, d)).astype(np.float32)).to(dev())
        e = rng.integers(0, n_ent, n)
        for r in (None, rng.integers(0, n_rel, n)):
            mats.requires_grad_(True), tab.requires_grad_(True)
            mats.grad = tab.grad = None
            r_t = None if r is None else torch.from_numpy(r).to(dev())
            out = ops.project_rows(mats, tab, r_t, n_rel - 1, torch.from_numpy(e).to(dev()))
            rr = np.full(n, n_rel - 1) if r is None else r
            M = mats.detach().cpu().numpy().astype(np.float64).reshape(n_rel, d, d)
            E = tab.detach().cpu().numpy().astype(np.float64)
            want = np.einsum("nck,nk->nc", M[rr], E[e])
            np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=2e-5, atol=2e-5)
            g = rng.standard_normal((n, d)).astype(np.float32)
            out.backward(torch.from_numpy(g).to(dev()))
            gE = np.zeros_like(E)
            np.add.at(gE, e, np.einsum("nck,nc->nk", M[rr], g.astype(np.float64)))
            gM = np.zeros_like(M)
            np.add.at(gM, rr, g.astype(np.float64)[:, :, None] * E[e][:, None, :])
            for got, ref in ((tab.grad, gE), (mats.grad, gM.reshape(n_rel, d * d))):
                scale = np.abs(ref).max()
                np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=3e-6 * scale)
