"""The C restatement (oracle/kge_ref.c, used as bench.py's cpu_baseline) against the pinned numpy oracle."""

import numpy as np
import pytest

from oracle import cref, kge
from tests.helpers import random_tables


@pytest.mark.parametrize("model", kge.MODELS)
def test_c_oracle_matches_numpy_oracle(model):
    dim, nU, nE, nR, B, G = 200, 50, 400, 6, 33, 17
    tables = random_tables(model, nU, nE, nR, dim, seed=1)
    rng = np.random.default_rng(0)
    h = rng.integers(0, nE, B)
    r = rng.integers(0, nR, B)
    t = rng.integers(0, nE, (B, G))
    hn, rn, tn = kge.slots(model, "kg")
    got = cref.score_grouped(model, [tables[n] for n in hn], [tables[n] for n in rn], [tables[n] for n in tn], h, r, 0, t, 1.0)
    ref = kge.score_grouped(model, tables, "kg", h, r, t, 1.0)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)
    assert cref.threads() >= 1
