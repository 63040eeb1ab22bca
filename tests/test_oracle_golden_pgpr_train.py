"""oracle/pgpr_train.py against the reference's recorded REINFORCE episode (tests/golden/pgpr_train_*.npz):
replaying the recorded action masks, dropout masks and sampled actions must reproduce the trajectory, rewards,
log-probs / values / entropies, the four loss parts and the gradients of all eight policy tensors."""

import os

import numpy as np
import pytest

from oracle import pgpr as op
from oracle import pgpr_train as opt
from oracle.philox import philox_words

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_train_oracle(tag, tie_rule="numpy"):
    g = np.load(os.path.join(GOLDEN, "pgpr_graph.npz"))
    z = np.load(os.path.join(GOLDEN, f"pgpr_train_{tag}.npz"))
    ckg = op.CKG(int(g["n_users"]), int(g["n_entities"]), int(g["ui_relation"]), g["inter_users"], g["inter_items"],
                 g["heads"], g["rels"], g["tails"])
    weights = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    orc = opt.PGPRTrainOracle(ckg, int(g["n_items"]), g["user_emb"], g["entity_emb"], g["relation_emb"], g["positives"],
                              weights, int(z["max_acts"]), 1, False, (10,), tie_rule=tie_rule, patterns=z["patterns"],
                              gamma=float(z["gamma"]), weight_factor=float(z["weight_factor"]))
    return g, z, orc


def replay(z, orc):
    users = z["users"][z["users"] != 0]
    T = z["acts"].shape[0]
    drops = [(z[f"drop1.{t}"], z[f"drop2.{t}"]) for t in range(T)]
    return orc.episode(users, lambda t, probs, mask: z["acts"][t], masks=z["masks"], drops=drops)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_episode_and_update(tag):
    g, z, orc = load_train_oracle(tag)
    steps, paths = replay(z, orc)
    want_paths = [[(int(r), int(t), int(n)) for r, t, n in p] for p in z["final_paths"]]
    assert paths == want_paths
    for t, s in enumerate(steps):
        np.testing.assert_array_equal(s["reward"], z["rewards"][t])
        np.testing.assert_allclose(s["logp"], z["logp"][t], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(s["value"], z["value"][t], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(s["entropy"], z["entropy"][t], rtol=2e-5, atol=2e-6)
        # the reference's dropout keeps action 0 and int((n-1)(1-p)) of the others when n >= 5 (pgpr.py:384-396)
        n, kept = s["n_acts"], z["masks"][t].sum(1)
        p = float(z["action_dropout"])
        want = np.where((n >= 5) & (p > 0), 1 + ((n - 1) * (1.0 - p)).astype(np.int64), n)
        np.testing.assert_array_equal(kept, want)
    assert (z["rewards"] > 0).any()
    parts, grads = orc.update(steps)
    np.testing.assert_allclose(parts, z["loss_parts"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(parts[0], float(z["loss"]), rtol=1e-5)
    for k, gr in grads.items():
        want = z[f"g.{k}"]
        assert np.abs(gr - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), k


def test_philox_known_answer():
    # Random123 known-answer vector for philox4x32-10: counter = key = 0
    w = philox_words(0, 0, np.arange(4))
    assert [hex(int(x)) for x in w] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


def test_device_draw_definitions():
    rng = np.random.default_rng(0)
    n_acts = np.array([1, 4, 5, 9, 51, 51])
    m = opt.dropout_keep_mask(n_acts, 51, 0.5, 2020, 6)
    assert (m[:, 0] == 1).all() and m.sum(1).tolist() == [1, 4, 3, 5, 26, 26]
    assert (m[4] != m[5]).any()
    assert opt.dropout_keep_mask(n_acts, 51, 0.0, 2020, 6).sum(1).tolist() == n_acts.tolist()
    p = rng.random((4000, 7)).astype(np.float32)
    p[:, 3] = 0
    p /= p.sum(1, keepdims=True)
    acts, u = opt.inverse_cdf_draw(np.tile(p[:1], (4000, 1)), 11, 3)
    assert (acts != 3).all() and 0 <= u.min() and u.max() < 1
    freq = np.bincount(acts, minlength=7) / 4000
    assert np.abs(freq - p[0]).max() < 0.03
