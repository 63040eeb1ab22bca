"""oracle/tprec.py against the reference TPRec class run on its PGPR fixture (tests/golden/tprec.npz)."""

import os

import numpy as np

from oracle import pgpr as op
from oracle import tprec as ot

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(tie_rule="numpy"):
    g = np.load(os.path.join(GOLDEN, "pgpr_graph.npz"))
    z = np.load(os.path.join(GOLDEN, "tprec.npz"))
    ckg = op.CKG(int(g["n_users"]), int(g["n_entities"]), int(g["ui_relation"]), g["inter_users"], g["inter_items"],
                 g["heads"], g["rels"], g["tails"])
    weights = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    ucw = ot.uc_weight_from_arrays(z["uc.users"], z["uc.clusters"], z["uc.weights"])
    orc = ot.TPRecOracle(ckg, int(g["n_items"]), z["user_emb"], z["entity_emb"], z["relation_emb"], g["positives"], weights,
                         int(z["max_acts"]), 1, False, (10,), tie_rule=tie_rule, patterns=(), gamma=float(z["gamma"]),
                         weight_factor=float(z["weight_factor"]), uc_weight=ucw, clust_emb=z["clust_emb"])
    return g, z, orc, ucw


def test_actions_with_temporal_vector():
    g, z, orc, _ = load()
    orc.start_user_revisitable = False  # beam_search converts the ids with .item() (tprec.py:658)
    flat, off = z["prim.paths_flat"], z["prim.paths_off"]
    aflat, aoff = z["prim.acts_flat"], z["prim.acts_off"]
    n_trim = 0
    for i in range(len(off) - 1):
        path = [(int(r), int(t), int(n)) for r, t, n in flat[off[i] : off[i + 1]]]
        want = [(int(r), int(n)) for r, n in aflat[aoff[i] : aoff[i + 1]]]
        got = orc.get_actions(path)
        if got != want:  # float64 ties at the trimming boundary only
            cand, sc = orc.action_scores(path)
            kth = np.sort(sc)[-orc.max_acts]
            assert all(sc[cand.index(a)] == kth for a in set(got) ^ set(want)), i
        n_trim += len(want) == orc.max_acts + 1
    assert n_trim > 5


def test_beam_and_collect():
    g, z, orc, _ = load()
    orc.start_user_revisitable = False
    np.testing.assert_allclose(orc.const, z["interacted_row"], rtol=1e-6)
    paths, probs, _ = orc.beam_search(z["users"], hops=(12, 4, 2))
    want = [[(int(r), int(t), int(n)) for r, t, n in p] for p in z["beam.paths"]]
    assert paths == want
    np.testing.assert_allclose(np.array(probs, np.float32), z["beam.probs"], rtol=2e-5, atol=1e-7)
    scores, expl = orc.collect_scores(z["users"], paths, [list(p) for p in z["beam.probs"]])
    np.testing.assert_allclose(scores, z["collect.scores"], rtol=1e-6)
    assert [[e[0], e[1]] for e in expl] == z["collect.expl_uis"].tolist()


def test_pretrain_loss_and_cluster_gradient():
    g, z, orc, ucw = load()
    labels = {(int(u), int(i)): int(c) for u, i, c in zip(z["label.users"], z["label.items"], z["label.cluster"])}
    for tag in ("multi", "multi2"):
        inter = {k: z[f"pre.{tag}.{k}"] for k in ("user_id", "item_id", "neg_item_id", "head_id", "relation_id", "tail_id",
                                                    "neg_tail_id")}
        loss, gC = ot.pretrain_loss(z["user_emb"], z["entity_emb"], z["relation_emb"], z["pre.ui_clust_relation_embedding"],
                                    labels, inter, 1.0)
        np.testing.assert_allclose(loss, float(z[f"pre.{tag}.loss"]), rtol=1e-6)
        np.testing.assert_allclose(gC, z[f"pre.{tag}.grad"], rtol=1e-4, atol=1e-7)


def test_policy_episode_replay():
    g, z, orc, _ = load()
    users = z["train.users"][z["train.users"] != 0]
    T = z["train.acts"].shape[0]
    drops = [(z[f"train.drop1.{t}"], z[f"train.drop2.{t}"]) for t in range(T)]
    steps, paths = orc.episode(users, lambda t, p, m: z["train.acts"][t], masks=z["train.masks"], drops=drops)
    assert paths == [[(int(r), int(t), int(n)) for r, t, n in p] for p in z["train.final_paths"]]
    assert (z["train.rewards"] == 0).all() and all((s["reward"] == 0).all() for s in steps)
    parts, grads = orc.update(steps)
    np.testing.assert_allclose(parts, z["train.loss_parts"], rtol=1e-5, atol=1e-6)
    for k, gr in grads.items():
        want = z[f"train.g.{k}"]
        assert np.abs(gr - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), k
