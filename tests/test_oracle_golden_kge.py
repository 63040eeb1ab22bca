"""The numpy oracle (oracle/kge.py, oracle/sampler.py) against vectors produced by the reference itself."""

import glob
import os

import numpy as np
import pytest

from oracle import kge, sampler

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
KGE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "kge_[A-Z]*_d*.npz")))


def load(path):
    z = np.load(path)
    tables = {k[len("table.") : -len(".weight")]: z[k] for k in z.files if k.startswith("table.")}
    inter = {k[len("inter.") :]: z[k] for k in z.files if k.startswith("inter.")}
    model = os.path.basename(path).split("_")[1]
    return z, model, tables, inter


@pytest.mark.parametrize("path", KGE_FILES, ids=[os.path.basename(p)[4:-4] for p in KGE_FILES])
def test_kge_oracle_matches_reference(path):
    z, model, tables, inter = load(path)
    m = float(z["margin"])
    np.testing.assert_allclose(kge.predict(model, tables, inter, m), z["predict"], rtol=1e-5, atol=1e-6)
    if "predict_kg" in z.files:  # TransH defines no kg-side predictors
        np.testing.assert_allclose(kge.predict_kg(model, tables, inter, m), z["predict_kg"], rtol=1e-5, atol=1e-6)
    loss, grads = kge.calculate_loss(model, tables, inter, m, with_grad=True)
    np.testing.assert_allclose(loss, z["loss"], rtol=2e-6)
    for k, g in grads.items():
        np.testing.assert_allclose(g, z[f"grad.{k}.weight"], rtol=2e-4, atol=2e-7, err_msg=k)
    n_items = z["fs.scores"].shape[1]
    fs = kge.full_sort_predict(model, tables, z["fs.users"], n_items, m)
    np.testing.assert_allclose(fs, z["fs.scores"], rtol=1e-5, atol=1e-6)
    if "fskg.scores" in z.files:
        fskg = kge.full_sort_predict_kg(model, tables, z["fskg.heads"], z["fskg.rels"], m)
        np.testing.assert_allclose(fskg, z["fskg.scores"], rtol=1e-5, atol=1e-6)
    # evaluation protocol: mask then top-10.  The reference's scores are used as input so that the
    # index comparison is exact (ties inside the -inf block are excluded by construction: k < #finite).
    masked = kge.mask_scores(z["fs.scores"], (z["eval.history_rows"], z["eval.history_cols"]))
    np.testing.assert_array_equal(masked, z["eval.masked"])
    tv, ti = kge.topk_desc(masked, 10)
    finite = np.isfinite(z["eval.topk_values"])
    np.testing.assert_array_equal(tv[finite], z["eval.topk_values"][finite])
    np.testing.assert_array_equal(ti[finite], z["eval.topk_idx"][finite])


def test_table_names_match_reference_state_dict():
    for path in KGE_FILES:
        z, model, tables, _ = load(path)
        assert sorted(tables) == sorted(kge.table_names(model))


def test_sampler_oracle_stream_exact():
    z = np.load(os.path.join(GOLDEN, "sampler.npz"))
    heads, tails, n = z["heads"], z["tails"], int(z["entity_num"])
    for dist in ("uniform", "popularity"):
        s = sampler.make_kg_sampler(heads, tails, n, dist, alpha=0.75)
        np.random.seed(2024)
        keys = z[f"kg.{dist}.keys"]
        np.testing.assert_array_equal(s.sample_by_key_ids(keys, 1), z[f"kg.{dist}.num1"])
        np.testing.assert_array_equal(s.sample_by_key_ids(keys, 3), z[f"kg.{dist}.num3"])
        np.testing.assert_array_equal(s.sample_by_key_ids(z[f"kg.{dist}.same_keys"], 4), z[f"kg.{dist}.same_num4"])
    phases = ["train", "valid", "test"]
    pu = [z[f"rec.{p}.users"] for p in phases]
    pi = [z[f"rec.{p}.items"] for p in phases]
    for dist in ("uniform", "popularity"):
        ss = sampler.make_rec_samplers(pu, pi, int(z["rec.user_num"]), int(z["rec.item_num"]), dist, 1.0)
        for p, s in zip(phases, ss):
            np.random.seed(11)
            got = s.sample_by_key_ids(z[f"rec.{dist}.{p}.keys"], 5)
            np.testing.assert_array_equal(got, z[f"rec.{dist}.{p}.num5"])


def test_sampler_oracle_properties():
    z = np.load(os.path.join(GOLDEN, "sampler.npz"))
    heads, tails, n = z["heads"], z["tails"], int(z["entity_num"])
    s = sampler.make_kg_sampler(heads, tails, n)
    keys = heads[:40]
    np.random.seed(0)
    v = s.sample_by_key_ids(keys, 6)
    assert v.shape == (240,) and v.min() >= 1 and v.max() < n
    for j in range(6):
        for i, k in enumerate(keys):
            assert int(v[j * len(keys) + i]) not in s.used[int(k)]  # layout j*len(keys)+i, sampler.py:148-152
    with pytest.raises(ValueError):
        s.sample_by_key_ids([n + 3], 1)
