"""CAFE device inference vs the reference golden (its own fixture) and the numpy oracle."""

import os

import numpy as np
import pytest
import torch

from oracle import cafe as oc
from tests.helpers import FIELDS, FakeKGDataset
from tests.test_oracle_golden_cafe import check_collected, load

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


class IdentityRng:
    def permutation(self, n):
        return np.arange(n)


def build():
    from hopwise_amd.model import CAFE

    z, orc, rid, ui = load()
    ds = FakeKGDataset(int(z["n_users"]), int(z["n_items"]), int(z["n_entities"]), len(rid), int(z["ui_relation"]),
                       z["inter_users"], z["inter_items"], z["heads"], z["rels"], z["tails"], z["positives"],
                       z["user_emb"], z["entity_emb"], np.zeros((len(rid), z["user_emb"].shape[1]), np.float32))
    ds.ui_relation = ui
    ds.field2token_id = {"relation_id": dict(rid)}
    metapaths = [[[None, "user"]] + [[str(r), "entity"] for r in mp] for mp in z["metapaths"]]
    cfg = dict(FIELDS, device=dev(), path_constraint=metapaths, rank_weight=10.0, deep_module=True, use_dropout=True,
               topk_candidates=10, sample_size=500, topk_paths=int(z["topk_paths"]), max_user_trials=50,
               load_embeddings=True)
    m = CAFE(cfg, ds).to(dev())
    # relation names contain dots, which nn.Module.load_state_dict cannot route ("model.film.actor.film.fc1.weight"):
    # assign tensor by tensor instead
    rels = sorted(m.relation_info, key=len, reverse=True)
    with torch.no_grad():
        for k in z.files:
            if not k.startswith("p.model."):
                continue
            rest = k[len("p.model."):]
            rel = next(r for r in rels if rest.startswith(r + "."))
            obj = m._module(rel)
            *path, leaf = rest[len(rel) + 1:].split(".")
            for a in path:
                obj = getattr(obj, a)
            getattr(obj, leaf).copy_(torch.from_numpy(z[k]))
    m.eval()
    m.rng = IdentityRng()
    return z, orc, m


def test_modules_and_log_softmax():
    from hopwise_amd import pgpr_ops as P

    z, orc, m = build()
    ev, uv = torch.from_numpy(z["mod.inputs_e"]).to(dev()), torch.from_numpy(z["mod.inputs_u"]).to(dev())
    with torch.no_grad():
        for k in z.files:
            if k.startswith("mod.") and not k.startswith("mod.inputs"):
                np.testing.assert_allclose(m._module(k[4:])((ev, uv)).cpu().numpy(), z[k], rtol=1e-4, atol=1e-5)
    x = np.random.default_rng(0).standard_normal((5, 6238)).astype(np.float32) * 3
    got = P.log_softmax_rows(torch.from_numpy(x).to(dev())).cpu().numpy()
    ref = np.stack([oc.log_softmax(r) for r in x])
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_infer_paths_vs_reference():
    z, orc, m = build()
    pred = m._infer_paths(torch.from_numpy(z["users"]))
    keys, off = z["infer.keys"], z["infer.off"]
    n = 0
    for t, (u, mpid) in enumerate(keys):
        ref_p = z["infer.paths"][off[t] : off[t + 1]].tolist()
        ref_v = z["infer.vals"][off[t] : off[t + 1]]
        got = pred[int(u)][int(mpid)]
        assert [p for p, _ in got] == ref_p, (u, mpid)
        if ref_p:
            np.testing.assert_allclose(np.array([v for _, v in got], np.float32), ref_v, rtol=1e-4, atol=1e-5)
            n += 1
    assert n > 20


def test_run_program_vs_reference():
    z, orc, m = build()
    scores, col = m.run_program(torch.from_numpy(z["users"]))
    res = scores.numpy()
    finite = np.isfinite(z["run.scores"])
    assert np.array_equal(np.isfinite(res), finite)
    np.testing.assert_allclose(res[finite], z["run.scores"][finite], rtol=1e-4)
    check_collected(col, z)
    fs, paths = m.full_sort_predict({"user_id": torch.from_numpy(z["users"])})
    assert all(isinstance(p[3][1][0], (int, np.integer)) for p in paths)  # relation ids after convert_path_relations
