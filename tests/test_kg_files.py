"""Atomic-file ingestion against the reference's own fixtures and the expectations of its
tests/data/test_dataset.py:865-893 (TestKGDataset.test_kg_remap_id / test_kg_reverse_r)."""

import os

import numpy as np

from hopwise_amd.data.kg_files import KGFiles, factorize

ATOMIC = os.path.join(os.path.dirname(__file__), "golden", "atomic")


def test_factorize_is_first_appearance_order():
    codes, uniq = factorize(["b", "a", "b", "c", "a"])
    assert codes.tolist() == [0, 1, 0, 2, 1] and uniq.tolist() == ["b", "a", "c"]


def test_kg_remap_id():
    ds = KGFiles(ATOMIC, "kg_remap_id")
    assert (ds.token2id("item_id", ["ib", "ic", "id"]) == [1, 2, 3]).all()
    assert (ds.token2id("entity_id", ["eb", "ec", "ed", "ee", "ea"]) == [1, 2, 3, 4, 5]).all()
    assert (ds.inter_feat["user_id"] == [1, 2, 3]).all()
    assert (ds.inter_feat["item_id"] == [1, 2, 3]).all()
    assert (ds.head_entities == [1, 2, 3, 4]).all()
    assert (ds.tail_entities == [5, 1, 2, 3]).all()
    assert ds.field2token_id["relation_id"][ds.ui_relation] == ds.relation_num - 1  # UI relation is the last id


def test_kg_reverse_r():
    ds = KGFiles(ATOMIC, "kg_reverse_r", kg_reverse_r=True)
    assert (ds.token2id("relation_id", ["ra", "rb", "ra_r", "rb_r"]) == [1, 2, 5, 6]).all()
    assert ds.relation_num == 10
    n = 4
    assert len(ds.relations) == 2 * n and (ds.relations[n:] == ds.relations[:n] + 4).all()
    assert (ds.head_entities[n:] == ds.tail_entities[:n]).all() and (ds.tail_entities[n:] == ds.head_entities[:n]).all()
    mat, _, cnt = ds.history_item_matrix()
    assert mat.shape[0] == ds.user_num and cnt.sum() == len(ds.inter_feat["user_id"])
