"""ConvE on the device: the reference goldens (eval + train mode with recorded masks) and the convolution kernels
against a plain PyTorch fp32 reference of the same op on the CPU."""

import os

import numpy as np
import pytest
import torch

from oracle import conve as oc
from tests.helpers import FakeDataset, make_config
from tests.test_oracle_golden_conve import FILES, NE, NI, NR, NU, load, train_masks

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def build(z, p):
    from hopwise_amd.model import ConvE

    d = p["user_embedding.weight"].shape[1]
    cfg = make_config(dev(), d, 1.0, label_smoothing=0.1, input_dropout=0.2, hidden_dropout=0.3, feature_dropout=0.2,
                      embedding_shape=int(z["embedding_shape"]), hidden_size=int(z["hidden_size"]), use_bias=True)
    m = ConvE(cfg, FakeDataset(NU, NI, NE, NR)).to(dev())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return m


def check_grads(m, z, prefix):
    # Parameters that sit in front of a batch-statistics BatchNorm (a bias, or a scale) have gradients that are exactly or
    # nearly zero by invariance: sums of large cancelling terms.  Their absolute error follows the magnitudes flowing
    # through the network, so the absolute floor is tied to the largest gradient of the model, not to the tiny result.
    overall = max(np.abs(z[f"{prefix}.{k}"]).max() for k, _ in m.named_parameters())
    for k, prm in m.named_parameters():
        ref = z[f"{prefix}.{k}"]
        scale = max(np.abs(ref).max(), 0.5 * overall)
        np.testing.assert_allclose(prm.grad.cpu().numpy(), ref, rtol=3e-4, atol=3e-6 * scale + 1e-9, err_msg=k)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_against_reference_golden(path):
    z, p, inter_np = load(path)
    m = build(z, p)
    inter = {k: torch.from_numpy(v).to(dev()) for k, v in inter_np.items()}
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m.predict(inter).cpu().numpy(), z["eval.predict"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(m.predict_kg(inter).cpu().numpy(), z["eval.predict_kg"], rtol=1e-4, atol=1e-6)
        fs = m.full_sort_predict({"user_id": torch.from_numpy(z["eval.fs_users"]).to(dev())})
        np.testing.assert_allclose(fs.cpu().numpy(), z["eval.fs"], rtol=1e-4, atol=1e-6)
        with pytest.raises(AttributeError):
            m.full_sort_predict_kg(inter)
    m.zero_grad()
    loss = m.calculate_loss(inter)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["eval.loss"]), rtol=1e-5)
    check_grads(m, z, "eval.grad")
    # train mode: batch-statistics BatchNorm + the reference's recorded dropout / dropout2d masks
    m = build(z, p)
    m.train()
    masks = train_masks(z)
    m.forced_masks = {s: tuple(torch.from_numpy(x).to(dev()) for x in mk) for s, mk in zip(("rec", "kg"), masks)}
    m.zero_grad()
    loss = m.calculate_loss(inter)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(z["train.loss"]), rtol=1e-5)
    check_grads(m, z, "train.grad")
    for k, v in m.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), z[f"train.after.{k}"], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("B,C,H,W,kh,kw", [(1, 1, 3, 3, 3, 3), (7, 32, 40, 10, 3, 3), (33, 5, 9, 4, 2, 3), (64, 64, 50, 3, 1, 3),
                                            (300, 32, 8, 8, 3, 3), (5, 3, 12, 7, 5, 6)])
def test_conv_kernels_vs_torch(B, C, H, W, kh, kw):
    from hopwise_amd import ops

    rng = np.random.default_rng(B + C)
    x = torch.from_numpy(rng.standard_normal((B, 1, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((C, 1, kh, kw)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(C).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((B, C, H - kh + 1, W - kw + 1)).astype(np.float32))
    xr, wr, br = (t.clone().double().requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.conv2d(xr, wr, br)
    ref.backward(g.double())
    xd, wd, bd = (t.to(dev()).requires_grad_(True) for t in (x, w, b))
    out = ops.conv2d_1ch(xd, wd, bd)
    out.backward(g.to(dev()))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    for got, want in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        scale = want.abs().max().item()
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-6 * scale)
    out2 = ops.conv2d_1ch(xd.detach(), wd.detach(), None)
    np.testing.assert_allclose(out2.cpu().numpy(), (ref - br.view(1, -1, 1, 1)).detach().numpy(), rtol=1e-5, atol=1e-5)


def test_default_shape_forward_vs_oracle():
    """embedding_size 200 / embedding_shape 20 / hidden 9728 (ConvE.yaml) in eval mode against the numpy oracle."""
    from hopwise_amd.model import ConvE

    torch.manual_seed(0)
    cfg = make_config(dev(), 200, 1.0, label_smoothing=0.1, input_dropout=0.2, hidden_dropout=0.3, feature_dropout=0.2,
                      embedding_shape=20, hidden_size=9728, use_bias=True)
    m = ConvE(cfg, FakeDataset(NU, NI, NE, NR)).to(dev())
    m.eval()
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    users = np.array([1, 5, 0, 19, 7])
    with torch.no_grad():
        fs = m.full_sort_predict({"user_id": torch.from_numpy(users).to(dev())}).cpu().numpy()
    ref = oc.forward(p, users, np.full_like(users, NR - 1), "user_embedding.weight", "user_embedding.weight", "b_users", 20)
    np.testing.assert_allclose(fs, ref[:, NU:], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("shape", [(7, 1, 20, 20), (130, 32, 18, 18), (2048, 5, 3, 7), (1, 3, 4, 4)])
def test_batchnorm2d_kernels_match_torch_in_training_mode(shape):
    """ops.BatchNorm2d (csrc/batchnorm.hip) against nn.BatchNorm2d evaluated in float64: output, the three gradients,
    running statistics (unbiased variance) and num_batches_tracked after two steps."""
    from hopwise_amd import ops

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(sum(shape))
    ref = torch.nn.BatchNorm2d(shape[1]).double()
    mine = ops.BatchNorm2d(shape[1]).to(dev)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(shape[1], generator=g).double())
        ref.bias.copy_(torch.randn(shape[1], generator=g).double())
        mine.weight.copy_(ref.weight.float())
        mine.bias.copy_(ref.bias.float())
    for step in range(2):
        x = torch.randn(shape, generator=g) * 2 + 0.5
        w = torch.randn(shape, generator=g)
        if shape[0] * shape[2] * shape[3] == 1:
            continue
        xr = x.double().requires_grad_()
        xm = x.to(dev).requires_grad_()
        yr, ym = ref(xr), mine(xm)
        (yr * w.double()).sum().backward()
        (ym * w.to(dev)).sum().backward()
        np.testing.assert_allclose(ym.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=2e-5)
        scale = float(xr.grad.abs().max())
        np.testing.assert_allclose(xm.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(mine.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=2e-4, atol=1e-3)
        np.testing.assert_allclose(mine.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), rtol=2e-4, atol=1e-3)
        mine.zero_grad()
        ref.zero_grad()
    np.testing.assert_allclose(mine.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mine.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked)
    mine.eval()
    ref.eval()
    x = torch.randn(shape, generator=g)
    np.testing.assert_allclose(mine(x.to(dev)).detach().cpu().numpy(), ref(x.double()).detach().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("N,C", [(2048, 200), (7, 5), (130, 64), (33, 257)])
def test_batchnorm1d_and_dropout_modules(N, C):
    """ops.BatchNorm1d against nn.BatchNorm1d (training forward, gradients, running statistics; evaluation forward) and
    ops.Dropout's Philox mask (keep rate, scaling, backward = same mask, reproducible per call counter)."""
    import copy

    from hopwise_amd import ops
    from oracle.philox import philox_words

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(N + C)
    x = (torch.randn(N, C, generator=g) * 2 + 0.5).to(dev)
    ref = torch.nn.BatchNorm1d(C).to(dev)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.normal_()
    mine = ops.BatchNorm1d(C).to(dev)
    mine.load_state_dict(copy.deepcopy(ref.state_dict()))
    w = torch.randn(N, C, generator=g).to(dev)
    for m in (ref, mine):
        m.train()
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = ref(xa), mine(xb)
    np.testing.assert_allclose(yb.detach().cpu().numpy(), ya.detach().cpu().numpy(), rtol=1e-4, atol=2e-5)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    np.testing.assert_allclose(xb.grad.cpu().numpy(), xa.grad.cpu().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(mine.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(mine.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), rtol=2e-4, atol=2e-3)
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        np.testing.assert_allclose(getattr(mine, k).cpu().numpy(), getattr(ref, k).cpu().numpy(), rtol=1e-5, atol=1e-6)
    for m in (ref, mine):
        m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(mine(x).cpu().numpy(), ref(x).cpu().numpy(), rtol=1e-5, atol=1e-5)
    # dropout
    d = ops.Dropout(0.3, seed=11, stream=5).to(dev)
    d.train()
    xd = x.clone().requires_grad_()
    y = d(xd)
    words = philox_words(d._seed, 1, np.arange(N * C, dtype=np.uint64))
    keep = ((words >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0) >= np.float32(0.3)).reshape(N, C)
    np.testing.assert_allclose(y.detach().cpu().numpy(), np.where(keep, x.cpu().numpy() / np.float32(0.7), 0.0), rtol=1e-6)
    y.sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), np.where(keep, 1.0 / 0.7, 0.0).astype(np.float32), rtol=1e-6)
    y2 = d(x)
    assert not torch.equal(y2, y.detach())  # next call, next offset
    d.eval()
    assert torch.equal(d(x), x)
