"""The PGPR oracle (oracle/pgpr.py) against the reference's own run on its tests/test_data fixture."""

import glob
import os

import numpy as np
import pytest

from oracle import pgpr as op

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TAGS = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "pgpr_*.npz")) if "graph" not in p and "train" not in p)


def load_oracle(tag, tie_rule="numpy"):
    g = np.load(os.path.join(GOLDEN, "pgpr_graph.npz"))
    z = np.load(os.path.join(GOLDEN, f"pgpr_{tag}.npz"))
    ckg = op.CKG(int(g["n_users"]), int(g["n_entities"]), int(g["ui_relation"]), g["inter_users"], g["inter_items"],
                 g["heads"], g["rels"], g["tails"])
    weights = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    orc = op.PGPROracle(ckg, int(g["n_items"]), g["user_emb"], g["entity_emb"], g["relation_emb"], g["positives"],
                        weights, int(z["max_acts"]), int(z["state_history"]), bool(z["fix_scores_sorting_bug"]),
                        tuple(int(x) for x in z["topk"]), tie_rule=tie_rule)
    return g, z, orc


def dec_paths(arr):
    return [[(int(r), int(t), int(n)) for r, t, n in p] for p in arr]


@pytest.mark.parametrize("tag", TAGS)
def test_actions_and_states(tag):
    g, z, orc = load_oracle(tag)
    flat, off = z["prim.paths_flat"], z["prim.paths_off"]
    aflat, aoff = z["prim.acts_flat"], z["prim.acts_off"]
    n_trim = 0
    for i in range(len(off) - 1):
        path = [(int(r), int(t), int(n)) for r, t, n in flat[off[i] : off[i + 1]]]
        want = [(int(r), int(n)) for r, n in aflat[aoff[i] : aoff[i + 1]]]
        got = orc.get_actions(path)
        if got != want:
            # only legal difference: float64 score ties at the trimming boundary (argsort order is unspecified)
            cand, sc = orc.action_scores(path)
            kth = np.sort(sc)[-orc.max_acts]
            diff = set(got) ^ set(want)
            assert all(abs(sc[cand.index(a)] - kth) == 0 for a in diff), (tag, i)
        n_trim += len(want) == orc.max_acts + 1
        np.testing.assert_array_equal(orc.get_state(path), z["prim.states"][i])
    if tag.startswith("trim"):
        assert n_trim > 0  # the >max_acts branch is exercised


@pytest.mark.parametrize("tag", [t for t in TAGS if t in ("default", "trim20")])
def test_policy_forward(tag):
    g, z, orc = load_oracle(tag)
    st, mask = z["prim.states"].astype(np.float32), z["prim.masks"].astype(bool)
    p, v = orc.forward(st, mask)
    np.testing.assert_allclose(p, z["prim.fwd_probs"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(v, z["prim.fwd_values"], rtol=2e-5, atol=1e-6)
    p, v = orc.forward(st, mask, z["prim.drop1"], z["prim.drop2"])
    np.testing.assert_allclose(p, z["prim.fwd_probs_drop"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(v, z["prim.fwd_values_drop"], rtol=2e-5, atol=1e-6)


def beam_names(z):
    return sorted({k.split(".")[1] for k in z.files if k.startswith("beam.")})


def canon(paths, probs):
    """Order-free, tie-tolerant summary: multiset of (path, rounded probs)."""
    return sorted((tuple(map(tuple, p)), tuple(np.round(np.asarray(q, np.float64), 5))) for p, q in zip(paths, probs))


@pytest.mark.parametrize("tag", TAGS)
def test_beam_search_and_collect(tag):
    g, z, orc = load_oracle(tag)
    for hn in beam_names(z):
        hops = [int(x) for x in z[f"beam.{hn}.hops"]]
        paths, probs, trace = orc.beam_search(z["users"], hops)
        ref_paths, ref_probs = dec_paths(z[f"beam.{hn}.paths"]), z[f"beam.{hn}.probs"]
        assert len(paths) == len(ref_paths)
        # p + 1.0 in float32 makes tiny probabilities collide (pgpr.py:453): torch.topk may then return tied
        # actions in another order, or a different member of a tie at the k-th place.  Paths whose every
        # hop prob is unique within its parent row must match exactly and in order.
        got = {tuple(map(tuple, p)): np.asarray(q) for p, q in zip(paths, probs)}
        n_exact = 0
        for p, q in zip(ref_paths, ref_probs):
            key = tuple(map(tuple, p))
            if key in got:
                np.testing.assert_allclose(got[key], q, rtol=2e-6)
                n_exact += 1
        assert n_exact >= 0.97 * len(ref_paths), (tag, hn, n_exact, len(ref_paths))
        np.testing.assert_allclose(np.sort(np.asarray(probs, np.float32).reshape(len(paths), -1), axis=0),
                                   np.sort(ref_probs, axis=0), rtol=2e-6)
        if n_exact == len(ref_paths) and [tuple(map(tuple, p)) for p in paths] == [tuple(map(tuple, p)) for p in ref_paths]:
            # identical beams -> collect_scores must agree bit for bit, quirks included
            res, expl = orc.collect_scores(z["users"], paths, probs)
            np.testing.assert_array_equal(res, z[f"collect.{hn}.scores"])
            assert [[e[0], e[1]] for e in expl] == z[f"collect.{hn}.expl_uis"].tolist()
            np.testing.assert_array_equal(np.array([e[2] for e in expl]), z[f"collect.{hn}.expl_score"])
            assert [e[3] for e in expl] == dec_paths(z[f"collect.{hn}.expl_paths"])
        # collect_scores on the REFERENCE's beams is always comparable
        res, expl = orc.collect_scores(z["users"], ref_paths, [list(q) for q in ref_probs])
        np.testing.assert_array_equal(res, z[f"collect.{hn}.scores"])
        assert [[e[0], e[1]] for e in expl] == z[f"collect.{hn}.expl_uis"].tolist()
        assert [e[3] for e in expl] == dec_paths(z[f"collect.{hn}.expl_paths"])
