"""A8 / A18 against batches captured from the REFERENCE's own loaders (tests/golden/make_golden_dropin.py ran hopwise's
Config -> create_dataset -> data_preparation on its `tests/test_data/test` fixture).  CPU only: the layout statements of
`hopwise_amd.driver.KGBatcher` and the oracle's loss / evaluation protocol are checked on what
`KnowledgeBasedDataLoader.__next__` (knowledge_dataloader.py:137-145) and `FullSortRecEvalDataLoader` /
`FullSortLPEvalDataLoader` (general_dataloader.py:170-281) really yielded."""

import os

import numpy as np
import pytest
import torch

from hopwise_amd.driver import KGBatcher
from oracle import kge as okge
from oracle import sampler as osamp

ID_KEYS = ("user_id", "item_id", "neg_item_id", "head_id", "relation_id", "tail_id", "neg_tail_id")
K, BS = 4, 64  # train_neg_sample_args.sample_num and train_batch_size of the capture


@pytest.fixture(scope="module")
def cap(golden_dir):
    return np.load(os.path.join(golden_dir, "dropin_RotatE_testdata.npz"))


def _used(cap, name):
    ptr, ids = cap[f"rec.{name}.ptr"], cap[f"rec.{name}.ids"]
    return [set(ids[ptr[i] : ptr[i + 1]].tolist()) for i in range(len(ptr) - 1)]


class _S:
    def __init__(self, s):
        self.s = s

    def sample_by_user_ids(self, users, items, num):
        return torch.from_numpy(self.s.sample_by_key_ids(np.asarray(users), num))

    def sample_by_entity_ids(self, heads, num=1):
        return torch.from_numpy(self.s.sample_by_key_ids(np.asarray(heads), num))


def test_reference_classes(cap):
    assert cap["rec.loader_classes"].tolist() == ["KnowledgeBasedDataset", "KnowledgeBasedDataLoader", "KGTrainer"]
    nU, nI, nE, nR, ui = cap["rec.stats"].tolist()
    assert (nU, nI, nE, nR) == (201, 301, 6136, 26) and ui == nR - 1  # SURVEY §4 fixture sizes; [UI-Relation] last


def test_reference_batch_has_the_layout_the_batcher_states(cap):
    nU, nI, nE, nR, ui = cap["rec.stats"].tolist()
    used_u, used_h = _used(cap, "rec_used"), _used(cap, "kg_used")
    P = BS // K
    pairs = set(zip(cap["rec.train.users"].tolist(), cap["rec.train.items"].tolist()))
    triples = set(zip(cap["rec.kg.heads"].tolist(), cap["rec.kg.rels"].tolist(), cap["rec.kg.tails"].tolist()))
    for b in range(2):
        keys = cap[f"rec.train{b}.keys"].tolist()
        assert set(ID_KEYS) <= set(keys)  # + joined user / item feature columns and their neg_ twins (ignored by KGE models)
        u, i, n = (cap[f"rec.train{b}.{k}"] for k in ID_KEYS[:3])
        assert u.dtype == i.dtype == n.dtype == np.int64 and len(u) == len(i) == len(n) == P * K
        for j in range(K):  # tile order: negative j of positive r sits at row j*P + r
            assert np.array_equal(u[j * P : (j + 1) * P], u[:P]) and np.array_equal(i[j * P : (j + 1) * P], i[:P])
        for r in range(P * K):
            assert (int(u[r]), int(i[r])) in pairs
            assert 1 <= n[r] < nI and int(n[r]) not in used_u[int(u[r])]
        h, r_, t, nt = (cap[f"rec.train{b}.{k}"] for k in ID_KEYS[3:])
        assert len(h) == len(r_) == len(t) == len(nt) == BS and h.dtype == np.int64
        for hh, rr, tt, nn in zip(h.tolist(), r_.tolist(), t.tolist(), nt.tolist()):
            assert (hh, rr, tt) in triples and 1 <= nn < nE and nn not in used_h[hh]
    assert int(cap["rec.train.len"]) == (len(cap["rec.train.users"]) + P - 1) // P


def test_batcher_yields_what_the_reference_loader_yields(cap):
    """Same source arrays, same batch size / sample_num -> same keys (the id columns), shapes, dtypes, epoch length."""
    nU, nI, nE, nR, ui = cap["rec.stats"].tolist()
    rec = osamp.OracleSampler(_used(cap, "rec_used"), nI)
    kgs = osamp.OracleSampler(_used(cap, "kg_used"), nE)
    np.random.seed(1)
    b = KGBatcher(cap["rec.train.users"], cap["rec.train.items"], cap["rec.kg.heads"], cap["rec.kg.rels"],
                  cap["rec.kg.tails"], _S(rec), _S(kgs), train_batch_size=BS, sample_num=K, seed=3)
    assert len(b) == int(cap["rec.train.len"])
    first = next(iter(b))
    for k in ID_KEYS:
        ref = cap[f"rec.train0.{k}"]
        assert tuple(first[k].shape) == ref.shape and first[k].dtype == torch.int64, k
    P = BS // K
    u = first["user_id"].numpy()
    assert all(np.array_equal(u[j * P : (j + 1) * P], u[:P]) for j in range(K))


def test_oracle_loss_on_reference_batches(cap):
    nU, nI, nE, nR, ui = cap["rec.stats"].tolist()
    tables = {k[len("rec.table.") : -len(".weight")]: cap[k] for k in cap.files if k.startswith("rec.table.")}
    for b in range(2):
        inter = {k: cap[f"rec.train{b}.{k}"] for k in ID_KEYS}
        got = okge.calculate_loss("RotatE", tables, inter, 1.0, ui)
        np.testing.assert_allclose(got, cap[f"rec.train{b}.loss"], rtol=1e-5)


@pytest.mark.parametrize("tag", ["rec", "lp"])
def test_oracle_eval_protocol_on_reference_batches(cap, tag):
    nU, nI, nE, nR, ui = cap[f"{tag}.stats"].tolist()
    tables = {k[len(tag) + 7 : -len(".weight")]: cap[k] for k in cap.files if k.startswith(f"{tag}.table.")}
    for b in range(int(cap[f"{tag}.eval.batches"])):
        p = f"{tag}.eval{b}"
        if tag == "rec":
            scores = okge.full_sort_predict("RotatE", tables, cap[f"{p}.inter.user_id"], nI, 1.0, ui)
        else:
            scores = okge.full_sort_predict_kg("RotatE", tables, cap[f"{p}.inter.head_id"], cap[f"{p}.inter.relation_id"], 1.0)
        scores = scores.reshape(cap[f"{p}.scores"].shape)
        np.testing.assert_allclose(scores, cap[f"{p}.scores"], rtol=1e-4, atol=1e-5)
        masked = okge.mask_scores(cap[f"{p}.scores"].copy(), (cap[f"{p}.history_rows"], cap[f"{p}.history_cols"]))
        assert np.array_equal(masked, cap[f"{p}.masked"])
        val, idx = okge.topk_desc(masked, 10)
        assert np.array_equal(val, cap[f"{p}.topk_values"])
        # torch.topk's order among equal values is unspecified; the build's is (value desc, index asc)
        ref_idx = cap[f"{p}.topk_idx"]
        differs = idx != ref_idx
        assert not differs.any() or np.array_equal(np.take_along_axis(masked, idx, 1)[differs],
                                                   np.take_along_axis(masked, ref_idx, 1)[differs])
