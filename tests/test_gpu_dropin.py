"""A8 / A18 on the device, against what the REFERENCE's own caller produced on its `tests/test_data/test` fixture
(tests/golden/dropin_RotatE_testdata.npz, captured by tests/golden/make_golden_dropin.py through hopwise's
Config -> create_dataset -> data_preparation):

  * the RSKG batches of `KnowledgeBasedDataLoader` -- every key, including the joined feature columns a KGE model
    ignores -- go through `calculate_loss` exactly as `Trainer._train_epoch` hands them over (trainer.py:243-259);
  * the `FullSortRecEvalDataLoader` / `FullSortLPEvalDataLoader` tuples go through `KGTrainer._full_sort_batch_eval`'s
    protocol (trainer.py:716-735) and the collector's top-k (collector.py:173-183), unfused and fused.
"""

import os

import numpy as np
import pytest
import torch

from hopwise_amd import driver, ops
from hopwise_amd.data import Interaction
from tests.helpers import FakeDataset, load_tables, make_config

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-5


def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def cap(golden_dir):
    return np.load(os.path.join(golden_dir, "dropin_RotatE_testdata.npz"))


def _model(cap, tag, **extra):
    from hopwise_amd.model import RotatE

    nU, nI, nE, nR, ui = cap[f"{tag}.stats"].tolist()
    tables = {k[len(tag) + 7 : -len(".weight")]: cap[k] for k in cap.files if k.startswith(f"{tag}.table.")}
    ds = FakeDataset(nU, nI, nE, nR)
    assert ds.field2token_id["relation_id"][ds.ui_relation] == ui
    neg = {"distribution": "uniform", "sample_num": 4, "alpha": 1.0, "dynamic": False, "candidate_num": 0}  # as captured
    m = RotatE(make_config(dev(), 16, 1.0, train_neg_sample_args=neg, **extra), ds).to(dev())
    return load_tables(m, tables, dev()), (nU, nI, nE, nR)


@pytest.mark.parametrize("extra", [{}, {"fused_loss": False}, {"rec_tiled_layout": False}], ids=["default", "per-op", "untiled"])
def test_reference_loader_batches_through_calculate_loss(cap, extra):
    m, _ = _model(cap, "rec", **extra)
    m.train()
    for b in range(2):
        keys = cap[f"rec.train{b}.keys"].tolist()
        inter = Interaction({k: cap[f"rec.train{b}.{k}"] for k in keys}).to(dev())  # whole batch H2D, as trainer.py:243
        m.zero_grad()
        losses = m.calculate_loss(inter)
        loss = sum(losses) if isinstance(losses, tuple) else losses
        loss.backward()
        np.testing.assert_allclose(loss.item(), cap[f"rec.train{b}.loss"], rtol=1e-5)
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def _check_topk(tv, ti, masked_ref, ref_v, ref_i):
    full_sorted = -np.sort(-masked_ref, axis=1)
    k = ref_v.shape[1]
    for r in range(ref_v.shape[0]):
        for c in range(k):
            if not np.isfinite(ref_v[r, c]):
                continue
            np.testing.assert_allclose(tv[r, c], ref_v[r, c], rtol=RTOL, atol=ATOL)
            gap_prev = np.inf if c == 0 else full_sorted[r, c - 1] - full_sorted[r, c]
            gap_next = full_sorted[r, c] - full_sorted[r, c + 1] if c + 1 < full_sorted.shape[1] else np.inf
            if min(gap_prev, gap_next) > 4 * (ATOL + RTOL * abs(ref_v[r, c])):
                assert ti[r, c] == ref_i[r, c], (r, c)


@pytest.mark.parametrize("tag", ["rec", "lp"])
def test_reference_eval_batches_through_full_sort_protocol(cap, tag):
    m, (nU, nI, nE, nR) = _model(cap, tag)
    kg = tag == "lp"
    n_cols = nE if kg else nI
    for b in range(int(cap[f"{tag}.eval.batches"])):
        p = f"{tag}.eval{b}"
        inter = Interaction({k[len(p) + 7 :]: cap[k] for k in cap.files if k.startswith(p + ".inter.")}).to(dev())
        hist = (torch.from_numpy(cap[f"{p}.history_rows"]).to(dev()), torch.from_numpy(cap[f"{p}.history_cols"]).to(dev()))
        m.eval()
        with torch.no_grad():
            raw = (m.full_sort_predict_kg if kg else m.full_sort_predict)(inter).view(-1, n_cols)
        np.testing.assert_allclose(raw.cpu().numpy(), cap[f"{p}.scores"], rtol=RTOL, atol=ATOL)
        val, idx, masked = driver.full_sort_topk(m, inter, hist, k=10, kg=kg)
        masked = masked.cpu().numpy()
        assert np.array_equal(np.isneginf(masked), np.isneginf(cap[f"{p}.masked"]))
        np.testing.assert_allclose(masked[np.isfinite(masked)], cap[f"{p}.masked"][np.isfinite(masked)], rtol=RTOL, atol=ATOL)
        _check_topk(val.cpu().numpy(), idx.cpu().numpy(), cap[f"{p}.masked"], cap[f"{p}.topk_values"], cap[f"{p}.topk_idx"])
        # fused scorer + mask + top-k (never writes [B, N]): same answer as the unfused path, bit for bit
        fv, fi = m.full_sort_topk(inter, 10, hist, kg=kg)
        assert torch.equal(fv, val) and torch.equal(fi, idx)
        # recall of the reference's positives inside the reference's top-10 == inside ours (what the metrics consume)
        pos_u, pos_i = cap[f"{p}.positive_u"], cap[f"{p}.positive_i"]
        ref_hit = (cap[f"{p}.topk_idx"][pos_u] == pos_i[:, None]).any(1)
        our_hit = (idx.cpu().numpy()[pos_u] == pos_i[:, None]).any(1)
        assert ref_hit.sum() == our_hit.sum()


def test_chunked_predict_fallback_matches(cap):
    """trainer.py:719-729: a model without full_sort_predict is scored by repeat_interleave x predict; same matrix."""
    m, (nU, nI, nE, nR) = _model(cap, "rec")
    p = "rec.eval0"
    users = torch.from_numpy(cap[f"{p}.inter.user_id"]).to(dev())
    inter = Interaction({"user_id": users}).repeat_interleave(nI)
    inter.update(Interaction({"item_id": torch.arange(nI, device=dev()).repeat(len(users))}))
    with torch.no_grad():
        got = m.predict(inter).view(-1, nI)
    np.testing.assert_allclose(got.cpu().numpy(), cap[f"{p}.scores"], rtol=RTOL, atol=ATOL)
