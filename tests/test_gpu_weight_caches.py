"""Tensors derived from a weight and cached on (data_ptr, _version) -- CAFE's transposed module weights and struct table,
PGPR's first-layer slot tables and pre-split policy planes -- must be rebuilt after an optimizer step.

hopwise_amd.optim.FusedAdam writes parameters through raw pointers (hw_adam_step), which autograd's version counters do not
see by themselves; driver.graphed_trainer installs it for every HIP model whose learner is adam, PGPR and CAFE included.
Without the explicit version bump every validation after the first one scored the weights of the first one.

Each test: evaluate (fills the caches), one FusedAdam step, evaluate again, and compare with a FRESH model object that
carries the stepped weights (no cache to be stale)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fused_adam_step(model, seed, lr=0.05):
    from hopwise_amd.optim import FusedAdam

    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=lr)
    gen = torch.Generator().manual_seed(seed)
    versions = [p._version for p in params]
    for p in params:
        p.grad.copy_(torch.randn(p.shape, generator=gen).to(p.device))
    opt.mark_touched()
    opt.step()
    assert all(p._version > v for p, v in zip(params, versions)), "hw_adam_step changed the parameters: versions must move"
    return opt


def test_pgpr_policy_caches_follow_a_fused_adam_step():
    from tests.test_gpu_pgpr import build_model, dev
    from tests.test_gpu_pgpr_fused import _logits, _rows

    g, z, orc, m = build_model("default")
    paths, recs = _rows(z)
    n = len(recs)
    rec = torch.from_numpy(recs).to(dev())
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    before = {f: _logits(m, rec, n_rows, n, None, fused=f) for f in (True, False)}  # caches filled for both forms
    _fused_adam_step(m, seed=1)
    _, _, _, fresh = build_model("default")
    fresh.load_state_dict(m.state_dict())
    for fused in (True, False):
        got = _logits(m, rec, n_rows, n, None, fused=fused)
        want = _logits(fresh, rec, n_rows, n, None, fused=fused)
        assert np.abs(got - before[fused]).max() > 1e-2, "the step must change the logits"
        np.testing.assert_array_equal(got, want)


def test_pgpr_caches_follow_a_graph_replay_free_torch_adam_step_too():
    """torch's own optimizers bump the version by themselves: the same check with torch.optim.Adam as the control."""
    from tests.test_gpu_pgpr import build_model, dev
    from tests.test_gpu_pgpr_fused import _logits, _rows

    g, z, orc, m = build_model("default")
    paths, recs = _rows(z)
    n = len(recs)
    rec = torch.from_numpy(recs).to(dev())
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    before = _logits(m, rec, n_rows, n, None, fused=True)
    opt = torch.optim.Adam(m.parameters(), lr=0.05)
    gen = torch.Generator().manual_seed(2)
    for p in m.parameters():
        p.grad = torch.randn(p.shape, generator=gen).to(p.device)
    opt.step()
    _, _, _, fresh = build_model("default")
    fresh.load_state_dict(m.state_dict())
    got = _logits(m, rec, n_rows, n, None, fused=True)
    assert np.abs(got - before).max() > 1e-2
    np.testing.assert_array_equal(got, _logits(fresh, rec, n_rows, n, None, fused=True))


@pytest.mark.parametrize("mode", ["2", "1"])
def test_cafe_module_caches_follow_a_fused_adam_step(monkeypatch, mode):
    """HW_CAFE_MODULES=2 (one launch per level: cached transposed weights + struct table) and =1 (C++ loop: struct table)
    against =0 (module-by-module torch loop, no derived cache) on the SAME model after the step."""
    from tests.test_gpu_cafe import build

    z, orc, m = build()
    users = torch.from_numpy(z["users"])
    monkeypatch.setenv("HW_CAFE_MODULES", mode)
    s0 = m.run_program_device(users)[0].cpu()  # caches filled
    _fused_adam_step(m, seed=3, lr=0.02)
    m.eval()
    s1, rec1, _, _ = m.run_program_device(users)
    monkeypatch.setenv("HW_CAFE_MODULES", "0")
    s2, rec2, _, _ = m.run_program_device(users)
    s1, s2 = s1.cpu().nan_to_num(-1e30), s2.cpu().nan_to_num(-1e30)
    assert (s1 - s0.nan_to_num(-1e30)).abs().max() > 1e-3, "the step must change the scores"
    if mode == "1":  # same kernels in the same order as the torch loop
        assert torch.equal(s1, s2)
    else:  # fp32 FMA chains instead of the split GEMM: fp32 rounding of the module outputs
        torch.testing.assert_close(s1, s2, rtol=1e-4, atol=1e-4)
    assert torch.equal(rec1["count"].cpu(), rec2["count"].cpu())
