"""A training step replayed from a HIP graph while eager work (poisoned temporaries) runs between the replays.

Regression test for the replica-scratch corruption found in round 3: hw_kge_grad_grouped zeroed its relation
replica scratch with hipMemsetAsync; captured in a graph that is a memset NODE, and a replayed memset node ran out of order
with the kernel nodes around it -- about one process in five saw the [UI-Relation] row of RotatE's relation table turn NaN
after 67-77 replays.  The library now fills with a kernel (hw::fill_bytes_async) and holds no hipMemsetAsync call."""
import glob
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_has_no_memset_nodes():
    for f in glob.glob(os.path.join(ROOT, "hopwise_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "hopwise_amd", "csrc", "*.h")) \
            + glob.glob(os.path.join(ROOT, "hopwise_amd", "csrc", "*.cpp")):
        src = open(f).read()
        code = "\n".join(ln.split("//")[0] for ln in src.splitlines())
        assert "hipMemsetAsync(" not in code and "hipMemset(" not in code and "hipMemset2D" not in code, f


@pytest.mark.gpu
def test_fill_bytes_any_size_and_alignment_and_under_replay():
    """hw_fill_bytes: unaligned heads and tails, every small size; and -- what a memset node does NOT do on this stack
    (tools/memset_pattern_probe.py) -- the captured value on every replay, with eager fills of other values in between."""
    from hopwise_amd import ops

    dev = torch.device("cuda", 0)
    buf = torch.empty(4096, dtype=torch.uint8, device=dev)
    for off in (0, 1, 2, 3, 5, 16, 33):
        for n in list(range(0, 40)) + [255, 256, 1021, 2048]:
            buf.fill_(0x55)
            ops.fill_bytes_(buf[off : off + n], 0xA7)
            want = torch.full((4096,), 0x55, dtype=torch.uint8)
            want[off : off + n] = 0xA7
            assert torch.equal(buf.cpu(), want), (off, n)
    dst = torch.empty(51200, dtype=torch.uint8, device=dev)
    other = torch.empty(1 << 16, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.fill_bytes_(dst, 0)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.fill_bytes_(dst, 0)
    for r in range(20):
        ops.fill_bytes_(other[r : r + 777], 1 + r)
        dst.fill_(0x55)
        g.replay()
        assert not bool(dst.any()), r


@pytest.mark.gpu
def test_graph_replay_with_eager_work_between_replays_stays_finite_and_tracks_the_eager_loop():
    import hopwise_amd.model as M
    from hopwise_amd.driver import GraphedTrainStep
    from hopwise_amd.optim import FusedAdam
    from tests.helpers import FakeDataset, make_config

    dev = torch.device("cuda", 0)
    nU, nI, nE, nR, K, B, DIM = 6041, 2985, 13805, 13, 256, 2048, 200
    rng = np.random.default_rng(0)
    pos_n = B // K
    users, pos = rng.integers(1, nU, pos_n), rng.integers(1, nI, pos_n)
    it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B),
        head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
        neg_tail_id=rng.integers(1, nE, B)).items()}

    def fresh():
        torch.manual_seed(0)
        m = M.RotatE(make_config(dev, DIM, 1.0, train_neg_sample_args={"sample_num": K}), FakeDataset(nU, nI, nE, nR)).to(dev)
        return m, FusedAdam(m.parameters(), lr=1e-3)

    steps = 100
    m0, o0 = fresh()
    for _ in range(steps):
        o0.zero_grad(set_to_none=True)
        loss0 = m0.calculate_loss(it)
        loss0.backward()
        o0.step()
    was = torch.utils.deterministic.fill_uninitialized_memory, torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True  # torch.empty hands out NaN-filled memory from here on
    try:
        m1, o1 = fresh()
        g = GraphedTrainStep(m1, o1, it)
        for i in range(steps):
            loss1 = g(it)
            # eager reductions between the replays: their staging buffers come from the ordinary pool
            junk = torch.empty(1 << 16, device=dev)
            assert bool(torch.isnan(junk).all())
            for n, p in m1.named_parameters():
                assert bool(torch.isfinite(p).all()), (i, n)
            assert bool(torch.isfinite(loss1))
    finally:
        torch.utils.deterministic.fill_uninitialized_memory = was[0]
        torch.use_deterministic_algorithms(was[1])
    for (n, p0), (_, p1) in zip(m0.named_parameters(), m1.named_parameters()):
        # float atomics: the two loops differ by summation order only, which Adam's m / sqrt(v) amplifies on the few elements
        # whose gradient is near zero (measured after 100 steps of lr 1e-3, a walk of up to 0.1 per element: mean |difference|
        # 1.8e-4 on the entity table, one element of 2.8 M off by 1.2e-2) -- so the comparison is in aggregate; a row that took garbage once is off in every column
        diff = (p1.detach() - p0.detach()).abs()
        assert float(diff.mean()) < 2e-4, (n, float(diff.mean()))
        assert float((diff > 2e-2).float().mean()) < 1e-3, (n, float((diff > 2e-2).float().mean()))
    assert abs(float(loss1.detach()) - float(loss0.detach())) < 2e-3 * abs(float(loss0.detach()))


@pytest.mark.gpu
def test_fill_kernel_on_unaligned_and_odd_sized_buffers():
    """hw::fill_bytes_async through an entry point that fills a caller's buffer (hw_pgpr_init_paths zeroes probs [3 n]): a
    probs view that starts 4 bytes off a 16-byte boundary (the kernel's dword path) and lengths that are not multiples of
    four words (the tail of its 16-byte path); the words around the view must stay untouched."""
    from hopwise_amd import pgpr_ops as P

    dev = torch.device("cuda", 0)
    for n, off in ((1, 0), (5, 0), (7, 1), (33, 3), (1000, 1), (4096, 2)):
        users = torch.arange(1, n + 1, device=dev, dtype=torch.int64)
        paths = torch.zeros((n, P.ROW_INTS), dtype=torch.int32, device=dev)
        backing = torch.full((3 * n + 8,), 7.0, device=dev)
        probs = backing[off : off + 3 * n]
        n_rows = torch.zeros(1, dtype=torch.int32, device=dev)
        P.init_paths(users, paths, probs, n_rows)
        torch.cuda.synchronize()
        assert int(n_rows.item()) == n
        assert bool((backing[off : off + 3 * n] == 0).all()), (n, off)
        assert bool((backing[:off] == 7).all()) and bool((backing[off + 3 * n :] == 7).all()), (n, off)


@pytest.mark.gpu
def test_a_capture_with_a_memset_node_is_refused():
    """GraphedTrainStep inspects the captured graph: a hipMemsetAsync inside the step (here smuggled into calculate_loss)
    would replay a stale pattern from the second launch on, so the capture is refused instead of instantiated."""
    import ctypes as C

    import hopwise_amd.model as M
    from hopwise_amd.driver import GraphedTrainStep, count_memset_nodes
    from hopwise_amd.optim import FusedAdam
    from tests.helpers import FakeDataset, make_config

    dev = torch.device("cuda", 0)
    nU, nI, nE, nR, nb = 60, 40, 120, 6, 64
    torch.manual_seed(0)
    m = M.TransE(make_config(dev, 16, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    rng = np.random.default_rng(0)
    it = {k: torch.from_numpy(rng.integers(1, hi, nb)).to(dev) for k, hi in (
        ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1), ("tail_id", nE),
        ("neg_tail_id", nE))}
    good = GraphedTrainStep(m, FusedAdam(m.parameters(), lr=1e-3), it)
    assert count_memset_nodes(good.graph) == 0
    assert bool(torch.isfinite(good(it)))

    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    scratch = torch.empty(1024, device=dev)
    m2 = M.TransE(make_config(dev, 16, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    plain = m2.calculate_loss

    def with_memset(interaction):
        assert hip.hipMemsetAsync(scratch.data_ptr(), 0, 4096, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        return plain(interaction)

    m2.calculate_loss = with_memset
    with pytest.raises(ValueError, match="memset node"):
        GraphedTrainStep(m2, FusedAdam(m2.parameters(), lr=1e-3), it)
