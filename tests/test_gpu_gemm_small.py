"""The 64 x 64-tile kernel for short products on few tiles (csrc/gemm_small.h) against the 128-wide kernels it replaces there
(HW_GEMM_SMALL=0: the same three-way split, the same term order -- equal bits) and against float64."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _run(a, b, akc, bkc, bias, elu, m_rows, small):
    from hopwise_amd import ops

    old = os.environ.get("HW_GEMM_SMALL")
    os.environ["HW_GEMM_SMALL"] = "1" if small else "0"
    try:
        M = a.shape[0] if akc else a.shape[1]
        N = b.shape[0] if bkc else b.shape[1]
        out = torch.full((M, N), 7.0, device=dev())  # rows beyond a device-side row count must stay untouched
        ops.gemm(a, b, akc, bkc, bias=bias, elu=elu, out=out, m_rows=m_rows)
        return out
    finally:
        if old is None:
            os.environ.pop("HW_GEMM_SMALL", None)
        else:
            os.environ["HW_GEMM_SMALL"] = old


@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, False), (False, True)])
def test_small_tiles_equal_the_wide_kernels_and_float64(akc, bkc):
    rng = np.random.default_rng(5)
    # (M, N, K): CAFE's module layers at 512 paths and their gradients, PGPR's policy layers on a first hop, ragged edges, a
    # single chunk, K beyond two chunks of prefetch, K that is no multiple of 4 in a padded row
    shapes = [(512, 256, 200), (512, 200, 256), (512, 100, 200), (256, 200, 512), (64, 512, 400), (37, 251, 256),
              (130, 70, 61), (5, 4, 4), (1000, 256, 1024), (63, 65, 129)]
    for M, N, K in shapes:
        ka, kb = (K + 3) & ~3, (K + 3) & ~3
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K)).astype(np.float32)
        if akc:
            a = torch.zeros((M, ka), device=dev())[:, :K]
            a.copy_(torch.from_numpy(A))
        else:
            a = torch.zeros((K, (M + 3) & ~3), device=dev())[:, :M]
            a.copy_(torch.from_numpy(A.T))
        if bkc:
            b = torch.zeros((N, kb), device=dev())[:, :K]
            b.copy_(torch.from_numpy(B))
        else:
            b = torch.zeros((K, (N + 3) & ~3), device=dev())[:, :N]
            b.copy_(torch.from_numpy(B.T))
        if (not akc and M < 4) or (not bkc and N < 4):
            continue
        bias_np = rng.standard_normal(N).astype(np.float32)
        for bias, elu in ((None, False), (torch.from_numpy(bias_np).to(dev()), True), (torch.from_numpy(bias_np).to(dev()), "relu")):
            got = _run(a, b, akc, bkc, bias, elu, None, True)
            ref = _run(a, b, akc, bkc, bias, elu, None, False)
            if K < 1024:  # from K = 1024 on ops.gemm hands the wide kernels a split-K workspace: another summation order
                np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy(), err_msg=f"{M} {N} {K} {elu}")
            else:
                np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-4)
            want = A.astype(np.float64) @ B.astype(np.float64).T
            if bias is not None:
                want = want + bias_np
            if elu is True:
                want = np.where(want > 0, want, np.expm1(want))
            elif elu == "relu":
                want = np.maximum(want, 0)
            scale = np.sqrt(K)
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=5e-6 * scale)  # fp32 accumulation over K terms of size ~1


def test_device_side_row_count_and_untouched_rows():
    rng = np.random.default_rng(1)
    M, N, K = 300, 256, 512
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev())
    b = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).to(dev())
    for live in (0, 1, 64, 65, 299, 300, 5000):
        m_rows = torch.tensor([live], dtype=torch.int32, device=dev())
        got = _run(a, b, True, True, None, False, m_rows, True)
        n = min(live, M)
        full = _run(a, b, True, True, None, False, None, True)
        np.testing.assert_array_equal(got[:n].cpu().numpy(), full[:n].cpu().numpy())
        assert bool((got[n:] == 7.0).all())


def test_small_tiles_are_the_path_taken_and_the_gradients_follow():
    """linear_act forward + backward on a CAFE-sized layer: every product of it goes through the small tiles (same numbers as
    with them switched off), gradients equal torch's at float32 accuracy."""
    from hopwise_amd import ops

    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal((512, 200)).astype(np.float32)).to(dev()).requires_grad_()
    w = torch.from_numpy((rng.standard_normal((256, 200)) * 0.1).astype(np.float32)).to(dev()).requires_grad_()
    bvec = torch.from_numpy(rng.standard_normal(256).astype(np.float32)).to(dev()).requires_grad_()
    gy = torch.from_numpy(rng.standard_normal((512, 256)).astype(np.float32)).to(dev())
    res = {}
    for small in ("1", "0"):
        os.environ["HW_GEMM_SMALL"] = small
        for t in (x, w, bvec):
            t.grad = None
        y = ops.linear_act(x, w, bvec, 2)
        y.backward(gy)
        res[small] = [t.detach().clone() for t in (y, x.grad, w.grad, bvec.grad)]
    os.environ.pop("HW_GEMM_SMALL")
    for p, q in zip(res["1"], res["0"]):
        np.testing.assert_array_equal(p.cpu().numpy(), q.cpu().numpy())
    for t in (x, w, bvec):
        t.grad = None
    yt = torch.relu(torch.nn.functional.linear(x.double(), w.double(), bvec.double()))
    yt.backward(gy.double())
    for got, want in zip(res["1"], (yt, x.grad, w.grad, bvec.grad)):
        np.testing.assert_allclose(got.cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-4)
