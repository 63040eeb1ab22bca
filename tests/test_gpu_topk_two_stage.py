"""hw_topk_rows_ws (long rows in two stages: chunk top-k in registers, then a merge) against hw_topk_rows (one block per row,
k passes; pinned by the oracle in test_gpu_kge.py) bit for bit, ties / -inf / NaN / strided rows included, and against a
numpy lexsort restatement of the order (value desc, index asc)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lexsort_topk(a, k):
    idx = np.empty((a.shape[0], k), np.int64)
    for r in range(a.shape[0]):
        idx[r] = np.lexsort((np.arange(a.shape[1]), -a[r].astype(np.float64)))[:k]
    return idx


@pytest.mark.parametrize("rows,cols,ld,k", [
    (3, 8192, 8192, 10),        # two full chunks
    (4, 1024, 1024, 10),        # the first size the two-stage form takes (round 5; 8 192 before): a quarter of one chunk
    (9, 2985, 3008, 16),        # ml-1m-small's items in padded, 16-byte aligned rows (the float4 form of the chunk kernel)
    (9, 2985, 2985, 16),        # ... and unaligned
    (3, 4097, 4100, 32),        # one column into a second chunk, aligned padded rows
    (5, 8193, 8200, 1),         # one column into a third chunk, padded rows
    (2, 45539, 45539, 64),      # Yelp2018's item count, the largest k of the two-stage form
    (7, 100001, 100003, 10),
    (1, 12288, 12288, 37),
    (300, 9000, 9000, 10),
])
def test_two_stage_equals_one_stage(rows, cols, ld, k):
    from hopwise_amd import _lib, ops

    dev = torch.device("cuda", 0)
    assert _lib.load().hw_topk_rows_workspace_bytes(rows, cols, k) == rows * ((cols + 4095) // 4096) * k * 8
    rng = np.random.default_rng(rows * 1000 + k)
    a = rng.standard_normal((rows, ld)).astype(np.float32)
    a[:, : cols // 2] = np.round(a[:, : cols // 2] * 4) / 4  # heavy ties: the index order decides
    a[0, rng.integers(0, cols, cols // 3)] = -np.inf
    if rows > 1:
        a[1, :cols] = 0.5  # one value everywhere: the answer is columns 0..k-1
        a[1, cols:] = 9.0  # the padding beyond n_cols must not be looked at
    buf = torch.from_numpy(a).to(dev)
    s = buf[:, :cols]
    v2, i2 = ops.topk_rows(s, k)
    v1, i1 = ops.topk_rows(s, k, two_stage=False)
    np.testing.assert_array_equal(i2.cpu().numpy(), i1.cpu().numpy())
    np.testing.assert_array_equal(v2.cpu().numpy(), v1.cpu().numpy())
    want = _lexsort_topk(a[: min(rows, 8), :cols], k)
    np.testing.assert_array_equal(i2.cpu().numpy()[: want.shape[0]], want)
    if rows > 1:
        np.testing.assert_array_equal(i2[1].cpu().numpy(), np.arange(k))


def test_rows_that_run_out_and_nan():
    from hopwise_amd import ops

    dev = torch.device("cuda", 0)
    a = np.full((2, 20000), -np.inf, np.float32)
    a[0, [5, 19999, 4096, 8191]] = [1.0, 1.0, 3.0, np.nan]  # NaN ranks first (torch.topk's rule and the one-stage kernel's)
    s = torch.from_numpy(a).to(dev)
    v2, i2 = ops.topk_rows(s, 6)
    v1, i1 = ops.topk_rows(s, 6, two_stage=False)
    np.testing.assert_array_equal(i2.cpu().numpy(), i1.cpu().numpy())
    np.testing.assert_array_equal(v2.cpu().numpy(), v1.cpu().numpy())
    assert i2[0, :4].tolist() == [8191, 4096, 5, 19999]
    assert i2[0, 4:].tolist() == [0, 1] and i2[1].tolist() == [0, 1, 2, 3, 4, 5]  # -inf columns in index order


def test_small_workspace_is_refused():
    from hopwise_amd import _lib

    dev = torch.device("cuda", 0)
    lib = _lib.load()
    s = torch.zeros(2, 9000, device=dev)
    val = torch.empty(2, 4, device=dev)
    idx = torch.empty(2, 4, dtype=torch.int64, device=dev)
    ws = torch.empty(8, dtype=torch.int64, device=dev)
    rc = lib.hw_topk_rows_ws(s.data_ptr(), 2, 9000, 9000, 4, val.data_ptr(), idx.data_ptr(), ws.data_ptr(), 64, None)
    assert rc == -4
    # short rows: the call is hw_topk_rows and takes no workspace
    assert lib.hw_topk_rows_workspace_bytes(2, 1000, 4) == 0
    assert lib.hw_topk_rows_ws(s.data_ptr(), 2, 1000, 9000, 4, val.data_ptr(), idx.data_ptr(), None, 0, None) == 0
    torch.cuda.synchronize()
    assert idx[0].tolist() == [0, 1, 2, 3]
