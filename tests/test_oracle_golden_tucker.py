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


WIDE = os.path.join(GOLDEN, "tucker_wide_d200.npz")


def load_wide(path=WIDE):
    """The d = 200 golden (tests/golden/make_golden.py:gen_tucker_wide): the core tensor is regenerated from its stored seed
    (numpy's frozen legacy stream) and validated against the stored probe values and sum; the [B, d, d] dropout masks are
    unpacked from bits."""
    z = np.load(path)
    nU, nI, nE, nR, d = (int(v) for v in z["shape"])
    p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
    core = (np.random.RandomState(int(z["core.seed"])).uniform(-1.0, 1.0, (d, d, d)) * float(z["core.scale"])).astype(np.float32)
    np.testing.assert_array_equal(core[::40, ::40, ::40], z["core.probe"])
    assert abs(core.astype(np.float64).sum() - float(z["core.sum"])) < 1e-9
    p["weights"] = core
    inter = {k[len("inter."):]: z[k] for k in z.files if k.startswith("inter.")}
    masks = []
    for i in range(6):
        shape = tuple(int(v) for v in z[f"train.maskshape{i}"])
        n = int(np.prod(shape))
        bits = np.unpackbits(z[f"train.maskbits{i}"])[:n].reshape(shape)
        masks.append(bits.astype(np.float32) / np.float32(1.0 - float(z[f"train.maskp{i}"])))
    return z, p, inter, masks, (nU, nI, nE, nR, d)


def test_tucker_oracle_wide_d200():
    """BASELINE configs[4]'s width against the reference, at the north star's 1e-4."""
    z, p, inter, masks, (nU, nI, nE, nR, d) = load_wide()
    np.testing.assert_allclose(ot.predict(p, inter, nR - 1), z["eval.predict"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ot.predict_kg(p, inter), z["eval.predict_kg"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ot.full_sort_predict(p, z["eval.fs_users"], nU, nR - 1), z["eval.fs"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ot.calculate_loss(p, inter, nU, nI, nE, nR - 1, 0.1), z["eval.loss"], rtol=1e-5)
    mk = {"rec": tuple(masks[:3]), "kg": tuple(masks[3:])}
    np.testing.assert_allclose(ot.calculate_loss(p, inter, nU, nI, nE, nR - 1, 0.1, True, mk), z["train.loss"], rtol=1e-5)
