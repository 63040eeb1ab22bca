"""The full sort of the bilinear scorers (DistMult, ComplEx, Analogy) as ONE product on the MFMA GEMM against the gather
kernel's full sort (itself pinned by the reference goldens in test_gpu_kge.py / test_gpu_kge_ext.py) and against the oracle:
scores to float32 rounding of the reordered sums, ranked indices identical wherever neighbouring scores are further apart
than that, masked top-k of the GEMM form equal to mask + top-k of its own matrix bit for bit."""

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


@pytest.mark.parametrize("model", ["DistMult", "ComplEx", "Analogy"])
def test_gemm_full_sort_equals_the_gather_kernel_and_the_oracle(model):
    from hopwise_amd import ops

    for dim, nE, nI, B, k in ((64, 1003, 700, 3, 10), (200, 777, 300, 17, 50), (24, 200, 130, 130, 20), (8, 90, 40, 1, 5)):
        nU, nR = 50, 6
        tables = random_tables(model, nU, nE, nR, dim, seed=dim + B)
        m = build(model, tables, dim, nU, nI, nE, nR)
        assert m.BILINEAR and m.gemm_full_sort
        rng = np.random.default_rng(B)
        users_np, heads_np, rels_np = rng.integers(0, nU, B), rng.integers(0, nE, B), rng.integers(0, nR, B)
        users, heads, rels = (torch.from_numpy(x).to(dev()) for x in (users_np, heads_np, rels_np))
        hist = (torch.from_numpy(rng.integers(0, B, 40)).to(dev()), torch.from_numpy(rng.integers(0, nI, 40)).to(dev()))
        with torch.no_grad():
            for kg, inter, n in ((False, {"user_id": users}, nI), (True, {"head_id": heads, "relation_id": rels}, nE)):
                fn = m.full_sort_predict_kg if kg else m.full_sort_predict
                got = fn(inter).view(B, n)
                m.gemm_full_sort = False
                ker = fn(inter).view(B, n)
                kv, ki = m.full_sort_topk(inter, min(k, n), hist, kg=kg)
                m.gemm_full_sort = True
                scale = float(ker.abs().max())
                np.testing.assert_allclose(got.cpu().numpy(), ker.cpu().numpy(), rtol=2e-5, atol=2e-6 * scale)
                want = (okge.full_sort_predict_kg(model, tables, heads_np, rels_np, m.margin) if kg else
                        okge.full_sort_predict(model, tables, users_np, nI, m.margin, m._rec_rel_full_sort()))
                np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-5)
                # masked top-k of the GEMM form == mask + top-k of its own matrix
                full = got.clone().contiguous()
                ops.mask_scores_(full, hist)
                rv, ri = ops.topk_rows(full, min(k, n))
                gv, gi = m.full_sort_topk(inter, min(k, n), hist, kg=kg)
                np.testing.assert_array_equal(gi.cpu().numpy(), ri.cpu().numpy())
                np.testing.assert_array_equal(gv.cpu().numpy(), rv.cpu().numpy())
                # ... and the same ranking as the kernel's wherever its neighbouring values are not within rounding
                kvn, kin, gin = kv.cpu().numpy(), ki.cpu().numpy(), gi.cpu().numpy()
                gap_ok = np.ones_like(kin, bool)
                gap_ok[:, :-1] &= (kvn[:, :-1] - kvn[:, 1:]) > 1e-5 * scale
                gap_ok[:, 1:] &= (kvn[:, :-1] - kvn[:, 1:]) > 1e-5 * scale
                finite = np.isfinite(kvn)
                assert (kin == gin)[gap_ok & finite].all()


def test_cached_side_by_side_table_follows_the_weights():
    """ComplEx keeps a cached [N, 2 d] copy of (re | im) for the product: rebuilt after an optimizer step (FusedAdam bumps the
    parameters' versions) and after load_state_dict."""
    from hopwise_amd.optim import FusedAdam

    dim, nU, nI, nE, nR = 16, 30, 40, 90, 5
    m = build("ComplEx", random_tables("ComplEx", nU, nE, nR, dim, seed=1), dim, nU, nI, nE, nR)
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


@pytest.mark.parametrize("model", ["DistMult", "ComplEx", "Analogy"])
def test_bilinear_topk_with_the_exact_rerank_is_the_gather_paths(model):
    """``gemm_rerank: True`` on a bilinear scorer (off by default there: the product's own scores are fp32-accurate): the
    evaluation top-k of the product form re-scored by the gather scorer and ranked exactly -- the gather path's columns, and
    values that are predict()'s bit for bit."""
    for dim, nE, nI, B, k in ((64, 1003, 700, 5, 10), (24, 5000, 4600, 70, 20)):
        nU, nR = 50, 6
        tables = random_tables(model, nU, nE, nR, dim, seed=dim + B)
        m = build(model, tables, dim, nU, nI, nE, nR, gemm_rerank=True)
        assert m.gemm_rerank and m.GEMM_FORM == "bilinear"
        rng = np.random.default_rng(B)
        users = torch.from_numpy(rng.integers(0, nU, B)).to(dev())
        hist = (torch.from_numpy(rng.integers(0, B, 40)).to(dev()), torch.from_numpy(rng.integers(0, nI, 40)).to(dev()))
        inter = {"user_id": users}
        with torch.no_grad():
            gv, gi = m.full_sort_topk(inter, k, hist)
            m.gemm_full_sort = False
            kv, ki = m.full_sort_topk(inter, k, hist)
            m.gemm_full_sort = True
            flat = {"user_id": users.repeat_interleave(k), "item_id": gi.reshape(-1)}
            np.testing.assert_array_equal(m.predict(flat).view(B, k).cpu().numpy(), gv.cpu().numpy())
        kvn, kin, gin = kv.cpu().numpy(), ki.cpu().numpy(), gi.cpu().numpy()
        scale = float(np.abs(kvn).max())
        np.testing.assert_allclose(gv.cpu().numpy(), kvn, rtol=1e-5, atol=1e-6 * scale)
        gap_ok = np.ones_like(kin, bool)
        gaps = kvn[:, :-1] - kvn[:, 1:]
        gap_ok[:, :-1] &= gaps > 2e-6 * scale
        gap_ok[:, 1:] &= gaps > 2e-6 * scale
        assert (kin == gin)[gap_ok].all() and gap_ok.mean() > 0.8


@pytest.mark.parametrize("model", ["DistMult", "ComplEx", "Analogy"])
def test_wide_tables_take_the_ranking_product_and_return_the_gather_paths_topk(model):
    """Default configuration, a table wide enough for the stretch maxima to mean something (n >= 128 c): the bilinear scorers'
    evaluation top-k runs the three-term ranking product (plane images from 512 queries on) with mask and maxima in its
    epilogue, re-scores its candidates with the gather scorer and proves the set in the product's own domain -- columns and
    values are the gather kernel's, bit for bit; with gemm_rerank: False the product's own values come back."""
    dim, nU, nI, nE, nR, k = 64, 700, 4600, 5000, 6, 10
    rng = np.random.default_rng(3)
    tables = random_tables(model, nU, nE, nR, dim, seed=7)
    m = build(model, tables, dim, nU, nI, nE, nR)
    assert m.gemm_rerank == "auto" and m.GEMM_FORM == "bilinear"
    for B in (70, 600):
        users = torch.from_numpy(rng.integers(0, nU, B)).to(dev())
        hist = (torch.from_numpy(rng.integers(0, B, 900)).to(dev()), torch.from_numpy(rng.integers(0, nI, 900)).to(dev()))
        with torch.no_grad():
            gv, gi = m.full_sort_topk({"user_id": users}, k, hist)
            m.gemm_full_sort = False
            kv, ki = m.full_sort_topk({"user_id": users}, k, hist)
            m.gemm_full_sort = True
        np.testing.assert_array_equal(gi.cpu().numpy(), ki.cpu().numpy())
        np.testing.assert_array_equal(gv.cpu().numpy(), kv.cpu().numpy())
    plain = build(model, tables, dim, nU, nI, nE, nR, gemm_rerank=False)
    with torch.no_grad():
        pv, pi = plain.full_sort_topk({"user_id": users}, k, hist)
        full = plain.full_sort_predict({"user_id": users}).view(len(users), nI).clone()
    from hopwise_amd import ops
    ops.mask_scores_(full, hist)
    rv, ri = ops.topk_rows(full, k)
    np.testing.assert_array_equal(pi.cpu().numpy(), ri.cpu().numpy())
    np.testing.assert_array_equal(pv.cpu().numpy(), rv.cpu().numpy())
