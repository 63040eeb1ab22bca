"""Beam searches of one PGPR model running concurrently on several HIP streams (what a caller overlapping two evaluation
batches does, and bench_pgpr.py --streams 2): same children, values and item scores as the same searches one after another.

Until round 4 hw_pgpr_actions' trim queues and the split-K workspace of the small policy products were ONE cached buffer per
(device, size) in pgpr_ops, shared by whatever ran at the time; three concurrent 256-user searches ended in a memory access
fault.  Scratch is allocated per call now (stream-ordered by torch's allocator)."""

import numpy as np
import pytest
import torch

from tests.helpers import FakeKGDataset, pgpr_config
from tests.test_gpu_pgpr_random import _graph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_policy", [True, False])
def test_concurrent_searches_equal_sequential_ones(fused_policy):
    from hopwise_amd.model.knowledge_aware_recommender import PGPR

    dev = torch.device("cuda", 0)
    sizes = (600, 80, 200, 6, 12000, 1500)  # hub items far over max_acts: every hop queues rows for the trim kernels
    n_users, n_items, n_entities, n_rel, n_inter, n_kg = sizes
    dim, hops = 16, (24, 12, 6)
    rng, ui, iu, ii, heads, rels, tails = _graph(5, *sizes)
    U = rng.standard_normal((n_users, dim)) / np.sqrt(dim)
    E = rng.standard_normal((n_entities, dim)) / np.sqrt(dim)
    R = rng.standard_normal((n_rel, dim)) / np.sqrt(dim)
    positives = np.zeros((n_users, 2), np.int64)
    ds = FakeKGDataset(n_users, n_items, n_entities, n_rel, ui, iu, ii, heads, rels, tails, positives, U, E, R)
    torch.manual_seed(0)
    cfg = pgpr_config(dev, 60, 1, hops, False, (10,), "identity")
    cfg["fused_policy"] = fused_policy
    m = PGPR(cfg, ds).to(dev)
    batches = [torch.from_numpy(rng.integers(1, n_users, 96)).to(dev) for _ in range(6)]

    def search(users):  # no host synchronisation in here: the launches of consecutive searches must be free to overlap
        res = m.beam_search_device(users, list(hops))
        scores, _ = m.collect_scores_device(users, res, n_nodes=4)
        return res, scores

    def settle(res, scores):  # after a device synchronisation
        n = res.n
        return res.paths[:n].clone(), res.probs[:n].clone(), scores.clone()

    want = []
    for b in batches:
        out = search(b)
        torch.cuda.synchronize()
        want.append(settle(*out))
    assert all(w[0].shape[0] > 1000 for w in want)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for _ in range(3):
        got = [None] * len(batches)
        for i, b in enumerate(batches):
            with torch.cuda.stream(streams[i % 3]):
                got[i] = search(b)
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            for a, c in zip(settle(*g), w):
                assert torch.equal(a, c)


def test_full_sort_predict_and_explain_over_streams_equal_the_sequential_ones():
    """An evaluation batch larger than ``beam_user_chunk``: the chunks' searches dealt over three side streams
    (``search_streams``, the default) give the score matrix and the explanation lists of the same chunks one after another."""
    from hopwise_amd.model.knowledge_aware_recommender import PGPR

    dev = torch.device("cuda", 0)
    sizes = (500, 70, 160, 5, 9000, 1200)
    n_users, n_items, n_entities, n_rel, n_inter, n_kg = sizes
    dim, hops = 16, (16, 8, 4)
    rng, ui, iu, ii, heads, rels, tails = _graph(9, *sizes)
    U = rng.standard_normal((n_users, dim)) / np.sqrt(dim)
    E = rng.standard_normal((n_entities, dim)) / np.sqrt(dim)
    R = rng.standard_normal((n_rel, dim)) / np.sqrt(dim)
    positives = np.zeros((n_users, 3), np.int64)
    for u in range(1, n_users):
        positives[u, :2] = rng.integers(1, n_items, 2)
    ds = FakeKGDataset(n_users, n_items, n_entities, n_rel, ui, iu, ii, heads, rels, tails, positives, U, E, R)
    torch.manual_seed(1)
    cfg = pgpr_config(dev, 40, 1, hops, False, (10,), "identity")
    cfg["beam_user_chunk"] = 48
    m = PGPR(cfg, ds).to(dev)
    assert m.search_streams == 3 and m._users_per_search() == 48
    users = torch.from_numpy(rng.integers(1, n_users, 300)).to(dev)  # seven chunks, the last one ragged
    inter = {m.USER_ID: users}
    m.search_streams = 1
    want = m.full_sort_predict(inter).clone()
    want_scores, want_expl = m.explain(inter)
    m.search_streams = 3
    for _ in range(3):
        got = m.full_sort_predict(inter)
        # the caller's stream goes straight on with the result: no synchronisation in between
        assert torch.equal(got, want)
    got_scores, got_expl = m.explain(inter)
    assert torch.equal(got_scores, want_scores)
    assert got_expl == want_expl and len(want_expl) > 100
    # ids handed over on the host (what a dataloader without pinned staging does): same result
    assert torch.equal(m.full_sort_predict({m.USER_ID: users.cpu()}), want)
    # the side streams are not model state (hopwise checkpoints state_dict() + other_parameter(): trainer.py:295-305)
    assert not any(isinstance(v, (list, tuple)) and v and isinstance(v[0], torch.cuda.Stream) for v in vars(m).values())
    sd = m.state_dict()
    assert all(torch.is_tensor(v) for v in sd.values())


def test_cafe_calls_over_streams_equal_sequential_ones():
    """CAFE.run_program_device holds no state between calls either (bench.py --cafe-streams 2)."""
    from tests.test_gpu_cafe import build

    z, orc, m = build()
    rng = np.random.default_rng(3)
    n_users = int(z["n_users"])
    batches = [torch.from_numpy(rng.integers(1, n_users, 40)) for _ in range(6)]  # host ids: a call then never waits
    want = []
    for b in batches:
        scores, rec, _, _ = m.run_program_device(b)
        torch.cuda.synchronize()
        want.append((scores.clone(), {k: v.clone() for k, v in rec.items() if torch.is_tensor(v)}))
    streams = [torch.cuda.Stream(device=torch.device("cuda", 0)) for _ in range(3)]
    for _ in range(3):
        got = [None] * len(batches)
        for i, b in enumerate(batches):
            with torch.cuda.stream(streams[i % 3]):
                got[i] = m.run_program_device(b)
        torch.cuda.synchronize()
        for (scores, rec, _, _), (ws, wr) in zip(got, want):
            assert torch.equal(torch.nan_to_num(scores, nan=-1.0), torch.nan_to_num(ws, nan=-1.0))
            for k, v in wr.items():
                assert torch.equal(rec[k], v), k


def test_collector_on_64_and_more_slots_equals_smaller_batches():
    """From 64 slots on the collector scores the items as a tiled product in a launch of its own (every block of the one-slot
    form re-reads the item table); the sums are taken in the same order, so scores and selections are those of the same
    users collected in batches below 64."""
    from hopwise_amd.model.knowledge_aware_recommender import PGPR

    dev = torch.device("cuda", 0)
    sizes = (1400, 90, 220, 5, 20000, 1500)
    n_users, n_items, n_entities, n_rel, n_inter, n_kg = sizes
    dim, hops = 12, (6, 3, 2)
    rng, ui, iu, ii, heads, rels, tails = _graph(21, *sizes)
    U = rng.standard_normal((n_users, dim)) / np.sqrt(dim)
    E = rng.standard_normal((n_entities, dim)) / np.sqrt(dim)
    R = rng.standard_normal((n_rel, dim)) / np.sqrt(dim)
    positives = np.zeros((n_users, 2), np.int64)
    ds = FakeKGDataset(n_users, n_items, n_entities, n_rel, ui, iu, ii, heads, rels, tails, positives, U, E, R)
    torch.manual_seed(2)
    m = PGPR(pgpr_config(dev, 30, 1, hops, False, (10,), "identity"), ds).to(dev)
    users = torch.from_numpy(rng.permutation(np.arange(1, n_users))[:1100]).to(dev)  # distinct: no folding of copies
    res = m.beam_search_device(users, list(hops))
    scores, sel = m.collect_scores_device(users, res, n_nodes=4)
    row = sel["row"].cpu().numpy()
    got_rows = {int(u): scores[r].clone() for u, r in zip(users.cpu().numpy(), row) if r >= 0}
    got_sel = {int(u): (sel["item"][i].clone(), sel["score"][i].clone(), int(sel["count"][i]))
               for i, u in enumerate(users.cpu().numpy())}
    assert len(got_rows) > 800
    for lo in range(0, 1100, 50):
        part = users[lo : lo + 50]
        res_p = m.beam_search_device(part, list(hops))
        sc_p, sel_p = m.collect_scores_device(part, res_p, n_nodes=4)
        row_p = sel_p["row"].cpu().numpy()
        for i, u in enumerate(part.cpu().numpy()):
            item, score, count = got_sel[int(u)]
            assert count == int(sel_p["count"][i])
            assert torch.equal(item[:count], sel_p["item"][i][:count]) and torch.equal(score[:count], sel_p["score"][i][:count])
            if row_p[i] >= 0:
                assert torch.equal(got_rows[int(u)], sc_p[row_p[i]])


@pytest.mark.parametrize("fused_policy", [True, False])
def test_first_multi_stream_evaluation_after_an_optimizer_step_reads_finished_caches(fused_policy):
    """The COLD case (round-4 review): the first-layer slot tables and the pre-split policy planes are rebuilt lazily after
    every optimizer step.  Dealt over three side streams, the first chunk used to rebuild them on ITS stream while the chunks
    on the other two streams found the cache key already set and read tables that were still being written.  The caches are
    built on the caller's stream ahead of the fan-out now, and carry their build event (stream_cache.FencedValue).

    A model whose caches have never been built, then one whose weights an optimizer step has just moved: the multi-stream
    ``full_sort_predict`` runs FIRST and must equal, bit for bit, a deep copy evaluated on one stream."""
    import copy

    from hopwise_amd.model.knowledge_aware_recommender import PGPR
    from hopwise_amd.optim import FusedAdam

    dev = torch.device("cuda", 0)
    sizes = (500, 70, 160, 5, 9000, 1200)
    n_users, n_items, n_entities, n_rel, n_inter, n_kg = sizes
    dim, hops = 100, (16, 8, 4)  # d = 100: the slot tables are (users + entities) x 512 floats per slot -- a build of some length
    rng, ui, iu, ii, heads, rels, tails = _graph(11, *sizes)
    U = rng.standard_normal((n_users, dim)) / np.sqrt(dim)
    E = rng.standard_normal((n_entities, dim)) / np.sqrt(dim)
    R = rng.standard_normal((n_rel, dim)) / np.sqrt(dim)
    positives = np.zeros((n_users, 3), np.int64)
    for u in range(1, n_users):
        positives[u, :2] = rng.integers(1, n_items, 2)
    ds = FakeKGDataset(n_users, n_items, n_entities, n_rel, ui, iu, ii, heads, rels, tails, positives, U, E, R)
    torch.manual_seed(2)
    cfg = pgpr_config(dev, 40, 1, hops, False, (10,), "identity")
    cfg["beam_user_chunk"] = 32
    cfg["fused_policy"] = fused_policy
    m = PGPR(cfg, ds).to(dev)
    users = torch.from_numpy(rng.integers(1, n_users, 320)).to(dev)  # ten chunks over three streams
    inter = {m.USER_ID: users}

    ref_model = PGPR(cfg, ds).to(dev)
    ref_model.search_streams = 1

    def sequential_copy():  # same weights, own caches, one stream
        ref_model.load_state_dict(copy.deepcopy(m.state_dict()))
        torch.cuda.synchronize()
        return ref_model

    # 1. never evaluated before: nothing is cached
    assert m.search_streams == 3 and "_l1_cache" not in vars(m) and "_fused_cache" not in vars(m)
    ref = sequential_copy()
    got = m.full_sort_predict(inter)
    want = ref.full_sort_predict(inter)
    assert torch.equal(got, want)
    assert torch.isfinite(got).any()
    first_expl = list(ref.explain(inter)[1])

    # 2. after optimizer steps (FusedAdam writes through raw pointers and bumps the version counters: the caches are stale)
    opt = FusedAdam(m.parameters(), lr=5e-2)
    for step in range(3):
        for p in m.parameters():  # FusedAdam's gradient sinks are the persistent p.grad buffers: written in place
            p.grad.copy_(torch.randn_like(p))
        opt.mark_touched()
        opt.step()
        ref = sequential_copy()
        got = m.full_sort_predict(inter)  # multi-stream FIRST, on cold caches
        want_step = ref.full_sort_predict(inter)
        assert torch.equal(got, want_step), f"step {step}"
        got_scores, got_expl = m.explain(inter)
        want_scores, want_expl = ref.explain(inter)
        assert torch.equal(got_scores, want_scores) and list(got_expl) == list(want_expl)
    assert list(want_expl) != first_expl  # the steps did move the policy: other paths are the best ones now


def test_fenced_value_orders_a_reader_on_another_stream_behind_the_build():
    """stream_cache.FencedValue by itself: a long build on one stream, a reader on another stream that finds the cached
    object must see the finished tensor."""
    from hopwise_amd.stream_cache import FencedValue

    dev = torch.device("cuda", 0)
    a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    src = torch.ones(64 << 20, device=dev)  # 256 MB
    torch.cuda.synchronize()
    fv = FencedValue()

    def build():
        t = src.clone()
        for _ in range(40):  # ~40 passes over 256 MB queued on stream a
            t.add_(1.0)
        return t

    with torch.cuda.stream(a):
        fv.get(1, build)
    with torch.cuda.stream(b):
        got = fv.get(1, build)[-1000:].clone()  # cache hit on the host at once; the device must wait for the adds
    torch.cuda.synchronize()
    assert torch.all(got == 41.0)
    import copy
    assert copy.deepcopy(fv).key is None
