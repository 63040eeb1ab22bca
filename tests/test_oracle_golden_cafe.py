"""CAFE oracle (oracle/cafe.py) against the reference's run on its own fixture."""

import os

import numpy as np
import pytest

from oracle import cafe as oc
from oracle.pgpr import CKG

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


TAGS = ("default", "shallow", "nodropout")  # CAFE.yaml; deep_module: False (RelationModule, cafe.py:645-667); use_dropout: False


def load(tag="default"):
    z = np.load(os.path.join(GOLDEN, f"cafe_{tag}.npz"))
    p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
    names = [str(x) for x in z["relation_names"]]
    rid = {n: i for i, n in enumerate(names)}
    ui_name = names[int(z["ui_relation"])]
    g = CKG(int(z["n_users"]), int(z["n_entities"]), int(z["ui_relation"]), z["inter_users"], z["inter_items"], z["heads"],
            z["rels"], z["tails"], ui_bidirectional=False)
    metapaths = [[(None, "user")] + [(str(r), "entity") for r in mp] for mp in z["metapaths"]]
    orc = oc.CafeOracle(g, p, z["user_emb"], z["entity_emb"], rid, ui_name, int(z["n_items"]), z["positives"], metapaths,
                        deep=bool(z["deep_module"]))
    return z, orc, rid, ui_name


def train_masks(z):
    """The multiplicative dropout masks of the recorded training step in call order (one per dropout call: two per level for
    the deep module, one for the shallow one; none recorded with use_dropout: False -> ones)."""
    n_levels, per = 3, 2 if bool(z["deep_module"]) else 1
    if not bool(z["use_dropout"]):
        return [np.float32(1.0)] * (n_levels * per)
    return [z[f"train.mask{i}"] for i in range(n_levels * per)]


@pytest.mark.parametrize("tag", TAGS)
def test_modules(tag):
    z, orc, rid, ui = load(tag)
    assert ("model." + str(z["metapaths"][0][0]) + ".fc3.weight" in orc.p) == bool(z["deep_module"])
    for k in z.files:
        if k.startswith("mod.") and not k.startswith("mod.inputs"):
            got = oc.module_forward(orc.p, k[4:], z["mod.inputs_e"], z["mod.inputs_u"], deep=bool(z["deep_module"]))
            np.testing.assert_allclose(got, z[k], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("tag", TAGS)
def test_infer_with_path(tag):
    z, orc, rid, ui = load(tag)
    keys, off = z["infer.keys"], z["infer.off"]
    n_nonempty = 0
    for t, (u, mpid) in enumerate(keys):
        ref_p = z["infer.paths"][off[t] : off[t + 1]].tolist()
        ref_v = z["infer.vals"][off[t] : off[t + 1]]
        got = orc.infer_with_path(orc.metapaths[mpid], int(u), z["positives"][u], int(z["topk_paths"]))
        assert [p for p, _ in got] == ref_p, (u, mpid)
        if ref_p:
            np.testing.assert_allclose(np.array([v for _, v in got], np.float32), ref_v, rtol=2e-5, atol=1e-5)
            n_nonempty += 1
    assert n_nonempty > 20


@pytest.mark.parametrize("tag", TAGS)
def test_run_program(tag):
    z, orc, rid, ui = load(tag)
    res, col = orc.run_program(z["users"])
    finite = np.isfinite(z["run.scores"])
    assert np.array_equal(np.isfinite(res), finite)
    np.testing.assert_allclose(res[finite], z["run.scores"][finite], rtol=3e-5)
    check_collected(col, z)


def check_collected(col, z):
    """Same (user, item) -> path map as the reference; the emission ORDER inside a user follows the float32 score sort,
    so it is only compared where neighbouring reference scores are further apart than the score tolerance."""
    ref = {(int(u), int(i)): (ids.tolist(), [str(r) for r in rels])
           for (u, i), ids, rels in zip(z["run.uis"], z["run.path_ids"], z["run.path_rels"])}
    got = {(c[0], c[1]): ([n[2] for n in c[3]], [n[0] for n in c[3][1:]]) for c in col}
    assert got == ref
    ref_order = [tuple(x) for x in z["run.uis"].tolist()]
    got_order = [(c[0], c[1]) for c in col]
    sc = z["run.score"]
    for k in range(len(ref_order)):
        near = any(0 <= j < len(sc) and ref_order[j][0] == ref_order[k][0] and abs(sc[j] - sc[k]) <= 1e-4 * abs(sc[k])
                   for j in (k - 1, k + 1))
        if not near:
            assert got_order[k] == ref_order[k], k


@pytest.mark.parametrize("tag", TAGS)
def test_training_forward_loss(tag):
    """SymbolicNetwork.forward in train mode on the reference's sampled batch (recorded dropout masks)."""
    z, orc, rid, ui = load(tag)
    mp = orc.metapaths[int(z["train.mpid"])]
    reg, rank = oc.symbolic_forward_loss(orc.p, mp, z["train.pos_paths"], z["train.neg_pids"],
                                         {"user": orc.U, "entity": orc.E}, train_masks(z), deep=bool(z["deep_module"]))
    np.testing.assert_allclose(reg, float(z["train.reg_loss"]), rtol=2e-5)
    np.testing.assert_allclose(rank, float(z["train.rank_loss"]), rtol=2e-5)
    # every sampled path is a walk of SOME metapath (the reference draws a metapath per user and scores the whole batch
    # with the last one drawn, cafe.py:161,198)
    for path in z["train.pos_paths"]:
        ok = False
        for cand in orc.metapaths:
            typ, cur, good = 0, int(path[0]), True
            for (rel, _), nxt in zip(cand[1:], path[1:]):
                rels, dsts = orc.g.neighbours(typ, cur)
                good = good and any(r == rid[rel] and d == nxt for r, d in zip(rels, dsts))
                typ, cur = 1, int(nxt)
            ok = ok or good
        assert ok, path
