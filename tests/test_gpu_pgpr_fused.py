"""The one-launch policy network of a beam-search hop (csrc/pgpr_policy.hip) against the three-launch form
(lookup -> l2 -> actor, each pinned by the reference goldens in tests/test_gpu_pgpr.py) and against a float64 numpy
restatement with the packed Philox dropout stream (oracle/philox.py:keep_bits_packed)."""

import numpy as np
import pytest
import torch

from tests.test_gpu_pgpr import build_model, dev, paths_to_records, prim_paths

pytestmark = pytest.mark.gpu


def _rows(z):
    paths = prim_paths(z)
    recs = paths_to_records(paths)
    roots = sorted({p[0][2] for p in paths})
    recs[:, 11] = [roots.index(p[0][2]) for p in paths]
    return paths, recs


def _logits(m, rec, n_rows, n, drops, fused):
    m.fused_policy = fused
    buf = dict(x1=torch.empty(n, m.hidden_sizes[0], device=dev()), x2=torch.empty(n, m.hidden_sizes[1], device=dev()),
               logits=torch.full((n, 256), 7.0, device=dev())[:, : m.act_dim])
    m._policy_logits(buf, n_rows, n, drops, 0, rec)
    return buf["logits"].cpu().numpy()


@pytest.mark.parametrize("tag", ["default", "hist0", "hist2", "trim20"])
def test_fused_policy_equals_three_launch_form(tag):
    """Identity dropout and explicit masks: same logits as lookup -> l2 -> actor up to the summation order of the products
    (the three-launch form is what the reference goldens pin)."""
    g, z, orc, m = build_model(tag)
    paths, recs = _rows(z)
    reps = 5  # a few hundred rows: more than one 128-row block, a ragged last block
    recs = np.tile(recs, (reps, 1))[: len(recs) * reps - 3]
    n = len(recs)
    rec = torch.from_numpy(recs).to(dev())
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    rng = np.random.default_rng(1)
    h1, h2 = m.hidden_sizes
    d1 = torch.from_numpy((rng.random((n, h1)) < 0.5).astype(np.float32) * 2).to(dev())
    d2 = torch.from_numpy((rng.random((n, h2)) < 0.5).astype(np.float32) * 2).to(dev())
    for drops in (None, [(d1, d2)]):
        want = _logits(m, rec, n_rows, n, drops, fused=False)
        got = _logits(m, rec, n_rows, n, drops, fused=True)
        scale = np.abs(want).max()
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6 * scale)


def test_fused_policy_row_count_on_device_and_untouched_tail():
    """n_rows is read on the device: rows past it are not written, whatever max_rows says."""
    g, z, orc, m = build_model("default")
    paths, recs = _rows(z)
    recs = np.tile(recs, (4, 1))
    cap = len(recs)
    n = cap - 130
    rec = torch.from_numpy(recs).to(dev())
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    got = _logits(m, rec, n_rows, cap, None, fused=True)
    want = _logits(m, rec, n_rows, cap, None, fused=False)
    np.testing.assert_allclose(got[:n], want[:n], rtol=2e-5, atol=2e-6 * np.abs(want[:n]).max())
    assert (got[n:] == 7.0).all()


def test_fused_policy_philox_dropout_stream():
    """drop_mode 3: the masks are the packed stream of oracle/philox.py (128 keep bits per Philox block), checked through a
    float64 restatement of the network on the kernel's own slot tables."""
    from oracle.philox import keep_bits_packed

    g, z, orc, m = build_model("default", dropout="reference")
    paths, recs = _rows(z)
    recs = np.tile(recs, (3, 1))
    n = len(recs)
    rec = torch.from_numpy(recs).to(dev())
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    m._drop_calls = 41
    got = _logits(m, rec, n_rows, n, None, fused=True)
    call = m._drop_calls  # advanced by _policy_logits
    h1, h2 = m.hidden_sizes
    k1 = keep_bits_packed(m._seed, 4 * call, n * h1).reshape(n, h1).astype(np.float64) * 2
    k2 = keep_bits_packed(m._seed, 4 * call + 1, n * h2).reshape(n, h2).astype(np.float64) * 2
    d1, d2 = torch.from_numpy(k1.astype(np.float32)).to(dev()), torch.from_numpy(k2.astype(np.float32)).to(dev())
    want = _logits(m, rec, n_rows, n, [(d1, d2)], fused=False)  # the goldens-pinned form with those masks handed in
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6 * np.abs(want).max())
    assert 0.4 < k1.mean() / 2 < 0.6 and 0.4 < k2.mean() / 2 < 0.6
