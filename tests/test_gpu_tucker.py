"""TuckER on the MFMA path vs the reference goldens (eval + train with recorded dropout masks) and the oracle."""

import glob
import os

import numpy as np
import pytest
import torch

from oracle import tucker as ot
from tests.helpers import FakeDataset, make_config
from tests.test_oracle_golden_tucker import FILES, NE, NI, NR, NU, load, load_wide

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def build(p, d, nU=NU, nI=NI, nE=NE, nR=NR):
    from hopwise_amd.model import TuckER

    cfg = make_config(dev(), d, 1.0, label_smoothing=0.1, input_dropout=0.3, input_dropout1=0.4, input_dropout2=0.5)
    m = TuckER(cfg, FakeDataset(nU, nI, nE, nR)).to(dev())
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    return m


def check_grads(m, z, prefix, rtol=3e-4):
    for k, prm in m.named_parameters():
        ref = z[f"{prefix}.grad.{k}"]
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(prm.grad.cpu().numpy(), ref, rtol=rtol, atol=3e-6 * scale + 1e-9, err_msg=k)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_tucker_vs_reference(path):
    z, p, inter = load(path)
    d = p["relation_embedding.weight"].shape[1]
    m = build(p, d)
    it = {k: torch.from_numpy(v).to(dev()) for k, v in inter.items()}
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m.predict(it).cpu().numpy(), z["eval.predict"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(m.predict_kg(it).cpu().numpy(), z["eval.predict_kg"], rtol=1e-4, atol=1e-6)
        fs = m.full_sort_predict({"user_id": torch.from_numpy(z["eval.fs_users"]).to(dev())})
        np.testing.assert_allclose(fs.cpu().numpy(), z["eval.fs"], rtol=1e-4, atol=1e-6)
    m.zero_grad()
    loss = m.calculate_loss(it)
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["eval.loss"], rtol=1e-5)
    check_grads(m, z, "eval")
    # one training-mode step with the reference's dropout masks
    m.train()
    t = lambda a: torch.from_numpy(a).to(dev())
    m.forced_masks = {"rec": (t(z["train.mask0"]), t(z["train.mask1"]), t(z["train.mask2"])),
                      "kg": (t(z["train.mask3"]), t(z["train.mask4"]), t(z["train.mask5"]))}
    m.zero_grad()
    loss = m.calculate_loss(it)
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["train.loss"], rtol=1e-5)
    check_grads(m, z, "train")
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), z[f"train.after.{k}"], rtol=1e-5, atol=1e-7, err_msg=k)


def test_tucker_vs_reference_wide_d200():
    """BASELINE configs[4]'s width (d = 200) against the REFERENCE's outputs at the north star's tolerance (1e-4 relative on
    the sigmoid outputs): eval-mode predict / predict_kg / full sort, loss and every gradient, and one training-mode step
    with the reference's recorded dropout masks (tests/golden/tucker_wide_d200.npz; the 32 MB core gradient is pinned by a
    [::5, ::5, ::5] sample plus float64 sums)."""
    z, p, inter, masks, (nU, nI, nE, nR, d) = load_wide()
    m = build(p, d, nU, nI, nE, nR)
    it = {k: torch.from_numpy(v).to(dev()) for k, v in inter.items()}

    def grads(prefix):
        overall = max(np.abs(z[f"{prefix}.grad.{k}"]).max() for k, _ in m.named_parameters() if k != "weights")
        for k, prm in m.named_parameters():
            g = prm.grad.cpu().numpy()
            if k == "weights":
                ref = z[f"{prefix}.grad.weights.sample"]
                scale = float(z[f"{prefix}.grad.weights.max_abs"])
                np.testing.assert_allclose(g[::5, ::5, ::5], ref, rtol=3e-4, atol=3e-6 * scale + 1e-9, err_msg=k)
                np.testing.assert_allclose(np.abs(g.astype(np.float64)).sum(), float(z[f"{prefix}.grad.weights.abs_sum"]), rtol=1e-5)
                np.testing.assert_allclose(g.astype(np.float64).sum(), float(z[f"{prefix}.grad.weights.sum"]),
                                           rtol=1e-4, atol=1e-6 * float(z[f"{prefix}.grad.weights.abs_sum"]))
            else:
                ref = z[f"{prefix}.grad.{k}"]
                scale = max(np.abs(ref).max(), 0.5 * overall)
                np.testing.assert_allclose(g, ref, rtol=3e-4, atol=3e-6 * scale + 1e-9, err_msg=k)

    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m.predict(it).cpu().numpy(), z["eval.predict"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(m.predict_kg(it).cpu().numpy(), z["eval.predict_kg"], rtol=1e-4, atol=1e-6)
        fs = m.full_sort_predict({"user_id": torch.from_numpy(z["eval.fs_users"]).to(dev())})
        np.testing.assert_allclose(fs.cpu().numpy(), z["eval.fs"], rtol=1e-4, atol=1e-6)
    m.zero_grad()
    loss = m.calculate_loss(it)
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["eval.loss"], rtol=1e-5)
    grads("eval")
    m.train()
    t = lambda a: torch.from_numpy(a).to(dev())
    m.forced_masks = {"rec": tuple(t(x) for x in masks[:3]), "kg": tuple(t(x) for x in masks[3:])}
    m.zero_grad()
    loss = m.calculate_loss(it)
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["train.loss"], rtol=1e-5)
    grads("train")
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), z[f"train.after.{k}"], rtol=1e-5, atol=1e-7, err_msg=k)


def test_tucker_vs_oracle_larger():
    """d = 200 (BASELINE config 5's width) on a ragged id space; eval-mode scores and loss vs the numpy oracle."""
    rng = np.random.default_rng(0)
    nU, nI, nE, nR, d, B = 300, 170, 900, 9, 200, 70
    from hopwise_amd.model import TuckER

    cfg = make_config(dev(), d, 1.0, label_smoothing=0.1, input_dropout=0.3, input_dropout1=0.4, input_dropout2=0.5)
    torch.manual_seed(1)
    m = TuckER(cfg, FakeDataset(nU, nI, nE, nR)).to(dev())
    with torch.no_grad():
        m.weights.mul_(0.05)
        m.bn0.running_mean.normal_(0, 0.1)
        m.bn1.running_var.uniform_(0.5, 2.0)
    p = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    inter = dict(user_id=rng.integers(1, nU, B), item_id=rng.integers(1, nI, B), head_id=rng.integers(1, nE, B),
                 relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B))
    it = {k: torch.from_numpy(v).to(dev()) for k, v in inter.items()}
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m.predict_kg(it).cpu().numpy(), ot.predict_kg(p, inter), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(m.full_sort_predict(it).cpu().numpy(), ot.full_sort_predict(p, inter["user_id"], nU, nR - 1),
                                   rtol=1e-4, atol=1e-6)
        loss = m.calculate_loss(it)
    np.testing.assert_allclose(loss.item(), ot.calculate_loss(p, inter, nU, nI, nE, nR - 1, 0.1), rtol=2e-5)
    # a training step runs (Philox-free torch dropout masks) and produces finite gradients everywhere
    m.train()
    m.zero_grad()
    m.calculate_loss(it).backward()
    assert all(torch.isfinite(q.grad).all() for q in m.parameters())


def test_contract_with_in_kernel_dropout():
    """Philox mode of the contraction: the mask is recovered with one-hot probes, then forward and both gradients must
    equal the explicit-mask path; keep rate and scale as configured."""
    from hopwise_amd import ops

    dev = torch.device("cuda", 0)
    for B, d, p in ((3, 8, 0.4), (5, 50, 0.25), (2, 200, 0.5)):
        torch.manual_seed(d)
        w1 = torch.ones(B, d, d, device=dev)
        mask = torch.empty(B, d, d, device=dev)
        for i in range(d):
            e = torch.zeros(B, d, device=dev)
            e[:, i] = 1
            mask[:, i, :] = ops.tucker_contract_dropout(e, w1, p, 77, 5)
        vals = torch.unique(mask).cpu().numpy()
        assert set(np.round(vals, 5)).issubset({0.0, np.float32(round(1 / (1 - p), 5))}), vals
        assert abs((mask > 0).float().mean().item() - (1 - p)) < 0.05
        x = torch.randn(B, d, device=dev, requires_grad=True)
        w = torch.randn(B, d, d, device=dev, requires_grad=True)
        go = torch.randn(B, d, device=dev)
        out = ops.tucker_contract_dropout(x, w, p, 77, 5)
        out.backward(go)
        gx, gw = x.grad.clone(), w.grad.clone()
        x.grad = w.grad = None
        ref = ops.tucker_contract(x, w, mask)
        ref.backward(go)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gx.cpu().numpy(), x.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(gw.cpu().numpy(), w.grad.cpu().numpy())
        other = ops.tucker_contract_dropout(x, w, p, 77, 6)  # another offset, another mask
        assert not torch.equal(other, out)


@pytest.mark.parametrize("shape", [(5, 37, 16), (300, 1001, 200), (129, 144, 36)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_score_loss_in_gemm_epilogue_equals_gemm_then_bce(shape, with_bias, monkeypatch):
    """hw_gemm_f32_bce (scores never stored: the GEMM epilogue writes dLoss/dscore and sums the loss) against the GEMM
    followed by the separate sigmoid + BCE sweep: same loss, same gradients of x / table / bias, ragged M and N."""
    from hopwise_amd import ops

    B, N, d = shape
    g = torch.Generator().manual_seed(B + N)
    x0 = torch.randn(B, d, generator=g)
    t0 = torch.randn(N, d, generator=g) * 0.3
    b0 = torch.randn(N, generator=g) * 0.1
    flag = (torch.rand(N, generator=g) < 0.05).to(torch.uint8).to(dev())
    out = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_SCORE_BCE", fused)
        x, t = x0.to(dev()).requires_grad_(), t0.to(dev()).requires_grad_()
        b = b0.to(dev()).requires_grad_() if with_bias else None
        loss = ops.score_all_bce_mean(x, t, flag, 0.9, 1.0 / N, bias=b)
        (2.0 * loss).backward()
        out.append((loss.item(), x.grad.cpu().numpy(), t.grad.cpu().numpy(), None if b is None else b.grad.cpu().numpy()))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-6)
    for a, c in zip(out[0][1:], out[1][1:]):
        if a is not None:
            np.testing.assert_allclose(a, c, rtol=2e-4, atol=1e-7 * max(1.0, float(np.abs(c).max())))


@pytest.mark.parametrize("d,B", [(200, 150), (24, 7), (64, 65)])
def test_shared_relation_matrices_equal_per_row_product(d, B):
    """The rec side with mm(r, W) computed ONCE ([UI-Relation] is the relation of every row) against the reference's
    per-row [B, d] x [d, d^2] product: same Philox dropout words, so the training forward is identical bit for bit and the
    gradients agree to summation order; also the explicit-mask and the eval paths."""
    import copy

    from hopwise_amd.model import TuckER

    rng = np.random.default_rng(d)
    nU, nI, nE, nR = 120, 70, 300, 7
    cfg = make_config(dev(), d, 1.0, label_smoothing=0.1, input_dropout=0.0, input_dropout1=0.4, input_dropout2=0.0)
    torch.manual_seed(3)
    a = TuckER(cfg, FakeDataset(nU, nI, nE, nR)).to(dev())
    with torch.no_grad():
        a.weights.mul_(0.05)
    b = copy.deepcopy(a)
    b.share_relation_matrices = False
    it = {k: torch.from_numpy(v).to(dev()) for k, v in dict(
        user_id=rng.integers(1, nU, B), item_id=rng.integers(1, nI, B), head_id=rng.integers(1, nE, B),
        relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B)).items()}
    for mode in ("train", "eval", "forced"):
        for m in (a, b):
            m.train(mode != "eval")
            m._drop_calls = 0
            m.forced_masks = None
            if mode == "forced":
                g = torch.Generator(device=dev()).manual_seed(5)
                mk = lambda *s: (torch.rand(*s, device=dev(), generator=g) > 0.3).float() / 0.7
                m.forced_masks = {"rec": (mk(B, d), mk(B, d, d), mk(B, d)), "kg": (mk(B, d), mk(B, d, d), mk(B, d))}
            m.zero_grad()
        la, lb = a.calculate_loss(it), b.calculate_loss(it)
        assert la.item() == lb.item(), mode
        la.backward()
        lb.backward()
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            ref = q.grad.cpu().numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=3e-4, atol=3e-6 * (np.abs(ref).max() + 1e-12) + 1e-9,
                                       err_msg=f"{mode}:{k}")


@pytest.mark.parametrize("d,B,n_mats", [(200, 33, 5), (24, 7, 1), (37, 65, 3)])
def test_kept_dropout_flags_give_the_regenerated_masks(d, B, n_mats):
    """hw_tucker_contract_shared_keep / _grad_keep (the forward stores its keep flags, the backward reads them) against
    the (seed, offset) entry points that draw the Philox blocks again: same output, same gx bit for bit; gw (float atomics
    per relation on the indexed side) to summation order."""
    import ctypes as C

    from hopwise_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(d)
    x = torch.randn(B, d, generator=g).to(dev())
    w = torch.randn(n_mats, d, d, generator=g).mul_(0.1).to(dev())
    index = torch.randint(0, n_mats, (B,), generator=g).to(dev()) if n_mats > 1 else None
    go = torch.randn(B, d, generator=g).to(dev())
    res = []
    for keep_path in (True, False):
        xs, ws = x.clone().requires_grad_(keep_path), w.clone().requires_grad_(keep_path)
        if keep_path:
            out = ops.tucker_contract_shared(xs, ws, index, None, 0.4, 11, 3)
            out.backward(go)
            res.append((out.detach(), xs.grad, ws.grad))
        else:  # no grad requested: the plain forward; the gradient through the regenerating entry point
            out = ops.tucker_contract_shared(xs, ws, index, None, 0.4, 11, 3)
            gx, gw = torch.empty_like(x), torch.empty_like(w)
            order = torch.argsort(index, stable=True) if index is not None else None
            n_ws = int(lib.hw_tucker_contract_shared_workspace_floats(B, d))
            wsp = torch.empty(n_ws, device=dev())
            ops.check(lib.hw_tucker_contract_shared_grad(ops._ptr(x), ops._ptr(w), ops._ptr(index), ops._ptr(order), n_mats, None,
                                                         ops._ptr(go), B, d, 0.4, C.c_uint64(11), C.c_uint64(3), ops._ptr(gx), ops._ptr(gw), ops._ptr(wsp),
                                                         n_ws, None), "hw_tucker_contract_shared_grad")
            res.append((out, gx, gw))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    if index is None:
        assert torch.equal(res[0][2], res[1][2])
    else:
        np.testing.assert_allclose(res[0][2].cpu().numpy(), res[1][2].cpu().numpy(), rtol=1e-5, atol=1e-6)
