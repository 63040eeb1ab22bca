"""oracle/convkb.py against the reference ConvKB class (tests/golden/convkb_d*.npz)."""

import glob
import os

import numpy as np
import pytest

from oracle import convkb as ok

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "convkb_d*.npz")))
NU, NI, NE, NR = 20, 16, 48, 5


def load(path):
    z = np.load(path)
    p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
    inter = {k[6:]: z[k] for k in z.files if k.startswith("inter.")}
    return z, p, inter


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_forward_and_losses(path):
    z, p, inter = load(path)
    U, E, R = p["user_embedding.weight"], p["entity_embedding.weight"], p["relation_embedding.weight"]
    ui = NR - 1
    got = ok.forward(p, U[inter["user_id"]], R[np.full_like(inter["user_id"], ui)], E[inter["item_id"]])
    np.testing.assert_allclose(got, z["eval.predict"], rtol=2e-5, atol=2e-6)
    got = ok.forward(p, E[inter["head_id"]], R[inter["relation_id"]], E[inter["tail_id"]])
    np.testing.assert_allclose(got, z["eval.predict_kg"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ok.calculate_loss(p, inter, ui, 0.2), float(z["eval.loss"]), rtol=1e-5)
    masks = [z[f"train.mask{i}"] for i in range(4)]
    np.testing.assert_allclose(ok.calculate_loss(p, inter, ui, 0.2, masks), float(z["train.loss"]), rtol=1e-5)
