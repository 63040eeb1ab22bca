"""TuckER oracle (oracle/tucker.py) against the reference's outputs."""

import glob
import os

import numpy as np
import pytest

from oracle import tucker as ot

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "tucker_d*.npz")))
NU, NI, NE, NR = 20, 16, 48, 5


def load(path):
    z = np.load(path)
    p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
    inter = {k[len("inter."):]: z[k] for k in z.files if k.startswith("inter.")}
    return z, p, inter


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_tucker_oracle(path):
    z, p, inter = load(path)
    np.testing.assert_allclose(ot.predict(p, inter, NR - 1), z["eval.predict"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ot.predict_kg(p, inter), z["eval.predict_kg"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ot.full_sort_predict(p, z["eval.fs_users"], NU, NR - 1), z["eval.fs"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ot.calculate_loss(p, inter, NU, NI, NE, NR - 1, 0.1), z["eval.loss"], rtol=1e-5)
    masks = {"rec": (z["train.mask0"], z["train.mask1"], z["train.mask2"]),
             "kg": (z["train.mask3"], z["train.mask4"], z["train.mask5"])}
    np.testing.assert_allclose(ot.calculate_loss(p, inter, NU, NI, NE, NR - 1, 0.1, True, masks), z["train.loss"], rtol=1e-5)
