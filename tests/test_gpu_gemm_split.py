"""fp32 GEMM (csrc/gemm.hip): the three-way bf16 split loop on the bf16 matrix pipes (default) and the exact-FMA loop
(HW_GEMM_SPLIT=0) against a float64 product, for the four operand layouts.  The error of a floating-point dot product is
bounded relative to sum_k |a_ik| |b_kj|, so that is the scale used; the split must not be worse than the FMA chain."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def padded(gen, r, c, wide):
    ld = (c + 31) // 32 * 32  # rows on 16-byte boundaries, as the engine allocates them: the pipelined loops run
    t = torch.randn((r, ld), device=dev(), generator=gen)
    if wide:  # five orders of magnitude of dynamic range
        t = t * torch.exp(3.0 * torch.randn((r, ld), device=dev(), generator=gen))
    return t[:, :c]


SHAPES = [(1, 7, 9), (3, 5, 200), (4, 4, 4), (130, 131, 33), (64, 200, 200), (257, 513, 72), (300, 1000, 40), (77, 256, 1234),
          (2048, 200, 5000)]


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, True), (False, False)])
def test_split_and_fma_loops_against_float64(monkeypatch, akc, bkc, wide):
    from hopwise_amd import ops

    gen = torch.Generator(device=dev()).manual_seed(7)
    worst = {"1": 0.0, "0": 0.0}
    for M, N, K in SHAPES:
        a = padded(gen, M, K, wide) if akc else padded(gen, K, M, wide)
        b = padded(gen, N, K, wide) if bkc else padded(gen, K, N, wide)
        A = a.double() if akc else a.double().t()
        Bm = b.double().t() if bkc else b.double()
        want, scale = A @ Bm, (A.abs() @ Bm.abs()).clamp_min(1e-300)
        for mode in ("1", "0"):
            monkeypatch.setenv("HW_GEMM_SPLIT", mode)
            got = ops.gemm(a, b, akc, bkc).double()
            worst[mode] = max(worst[mode], float(((got - want).abs() / scale).max()))
    assert worst["1"] < 3e-6 and worst["0"] < 3e-6, worst
    assert worst["1"] < 1.5 * worst["0"] + 1e-7, worst  # the split is as accurate as the fp32 FMA chain


def test_split_epilogues(monkeypatch):
    """bias + ELU / ReLU, Philox dropout (same keep bits in both loops), row-count pointer."""
    from hopwise_amd import ops

    gen = torch.Generator(device=dev()).manual_seed(9)
    a, b = padded(gen, 300, 400, False), padded(gen, 256, 400, False)
    bias = torch.randn(256, device=dev(), generator=gen)
    rows = torch.tensor([211], dtype=torch.int32, device=dev())
    for kw in (dict(bias=bias, elu=True), dict(bias=bias, elu="relu"), dict(bias=bias, elu=True, drop_mode=2, seed=5, offset=3),
               dict(bias=bias, m_rows=rows)):
        outs = []
        for mode in ("1", "0"):
            monkeypatch.setenv("HW_GEMM_SPLIT", mode)
            out = torch.full((300, 256), -7.0, device=dev())
            outs.append(ops.gemm(a, b, True, True, out=out, **kw))
        torch.testing.assert_close(outs[0], outs[1], rtol=1e-4, atol=1e-4)
        if "m_rows" in kw:
            assert bool((outs[0][211:] == -7.0).all())  # rows past the device-side count are not written
        if kw.get("drop_mode") == 2:
            assert torch.equal(outs[0] == 0, outs[1] == 0)


def test_split_exact_on_bf16_representable_integers(monkeypatch):
    """Small integers are exact in bf16, their products and sums exact in fp32: the split product equals the integer one."""
    from hopwise_amd import ops

    monkeypatch.setenv("HW_GEMM_SPLIT", "1")
    rng = np.random.default_rng(3)
    for akc, bkc in ((True, True), (True, False), (False, True), (False, False)):
        M, N, K = 150, 210, 96
        a = torch.zeros((M if akc else K, 128 if akc else 160), device=dev())[:, : (K if akc else M)]
        b = torch.zeros((N if bkc else K, 128 if bkc else 224), device=dev())[:, : (K if bkc else N)]
        a.copy_(torch.from_numpy(rng.integers(-9, 10, tuple(a.shape)).astype(np.float32)))
        b.copy_(torch.from_numpy(rng.integers(-9, 10, tuple(b.shape)).astype(np.float32)))
        A = a.double() if akc else a.double().t()
        Bm = b.double().t() if bkc else b.double()
        assert torch.equal(ops.gemm(a, b, akc, bkc).double(), A @ Bm), (akc, bkc)


@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, False)])
def test_wide_tiles_equal_the_128_wide_ones(monkeypatch, akc, bkc):
    """gemm_split_wide_kernel (128 x 256 tiles, plain products with 128 < N <= 256) against gemm_f32_kernel on the same
    operands: ragged M, N and K, a device-side row count, split-K over a long K, an output with padded rows."""
    from hopwise_amd import ops

    gen = torch.Generator(device=dev()).manual_seed(11)
    rows = torch.tensor([200], dtype=torch.int32, device=dev())
    for M, N, K, kw in ((257, 129, 50, {}), (300, 200, 203, dict(m_rows=rows)), (128, 256, 16, {}), (1000, 252, 4100, {}),
                        (1100, 200, 4200, {}), (2048, 130, 4096 + 5, dict(m_rows=torch.tensor([1999], dtype=torch.int32, device=dev())))):
        a = padded(gen, M, K, False) if akc else padded(gen, K, M, False)
        b = padded(gen, N, K, False) if bkc else padded(gen, K, N, False)
        outs = []
        for wide in ("1", "0"):
            monkeypatch.setenv("HW_GEMM_WIDE", wide)
            out = torch.full((M, (N + 31) // 32 * 32), -7.0, device=dev())[:, :N]
            outs.append(ops.gemm(a, b, akc, bkc, out=out, **kw))
        torch.testing.assert_close(outs[0], outs[1], rtol=2e-5, atol=2e-5 * float(outs[1].abs().max()))
        if "m_rows" in kw:
            assert bool((outs[0][int(kw["m_rows"][0]):] == -7.0).all())
