"""oracle/conve.py against the reference ConvE class (tests/golden/conve_d*.npz)."""

import glob
import os

import numpy as np
import pytest

from oracle import conve as oc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "conve_d*.npz")))
NU, NI, NE, NR = 20, 16, 48, 5


def load(path):
    z = np.load(path)
    p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
    inter = {k[6:]: z[k] for k in z.files if k.startswith("inter.")}
    return z, p, inter


def train_masks(z):
    # call order in calculate_loss: rec pass (inp_drop, feature_map_drop, hidden_drop), then the kg pass
    return ((z["train.mask0"], z["train.mask2d0"], z["train.mask1"]), (z["train.mask2"], z["train.mask2d1"], z["train.mask3"]))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_forward_and_losses(path):
    z, p, inter = load(path)
    shape = int(z["embedding_shape"])
    ui = NR - 1
    user, item = inter["user_id"], inter["item_id"]
    su = oc.forward(p, user, np.full_like(user, ui), "user_embedding.weight", "user_embedding.weight", "b_users", shape)
    np.testing.assert_allclose(su[:, NU:][np.arange(len(user)), item], z["eval.predict"], rtol=2e-5, atol=1e-6)
    sk = oc.forward(p, inter["head_id"], inter["relation_id"], "entity_embedding.weight", "entity_embedding.weight",
                    "b_entities", shape)
    np.testing.assert_allclose(sk[:, inter["tail_id"]], z["eval.predict_kg"], rtol=2e-5, atol=1e-6)
    fu = z["eval.fs_users"]
    fs = oc.forward(p, fu, np.full_like(fu, ui), "user_embedding.weight", "user_embedding.weight", "b_users", shape)[:, NU:]
    np.testing.assert_allclose(fs, z["eval.fs"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(oc.calculate_loss(p, inter, NU, NI, NE, ui, shape, 0.1), float(z["eval.loss"]), rtol=1e-5)
    np.testing.assert_allclose(oc.calculate_loss(p, inter, NU, NI, NE, ui, shape, 0.1, train_masks(z)),
                               float(z["train.loss"]), rtol=1e-5)
