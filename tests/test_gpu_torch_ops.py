"""torch.ops.hopwise_amd.* (csrc/torch_bind.cpp, TORCH_LIBRARY) against the reference goldens and against the ctypes
binding of the same C ABI: the registered ops, their autograd registration, and the one-op calculate_loss."""

import os

import numpy as np
import pytest
import torch

from tests.helpers import FakeDataset, load_tables, make_config, random_tables
from tests.test_gpu_kge import ATOL, KGE_FILES, RTOL, build, golden_case, to_dev

pytestmark = pytest.mark.gpu
GOLD = [p for p in KGE_FILES if os.path.basename(p).split("_")[1] in ("TransE", "RotatE", "ComplEx", "DistMult", "RESCAL")]


def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ns():
    from hopwise_amd import torch_ops

    return torch_ops.load()


def _tables(m, side):
    head, rel, tail = m._tables(side)
    return list(head), list(rel), list(tail)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[4:-4] for p in GOLD])
def test_registered_scorer_and_autograd_match_reference_golden(ns, path):
    from hopwise_amd import ops

    z, model, tables, inter = golden_case(path)
    d = next(iter(tables.values())).shape[1]
    nI = z["fs.scores"].shape[1]
    m = build(model, tables, d, 20, nI, 48, 5)
    it = to_dev(inter)
    mid = ops.MODEL_IDS[model]
    # predict (rotate.py:133-145): user, [UI-Relation], item
    head, rel, tail = _tables(m, "rec")
    got = ns.kge_score_grouped(mid, head, rel, tail, it["user_id"], None, m._rec_rel_loss(), it["item_id"].view(-1, 1),
                               m.margin, 0.0, False, 0)
    np.testing.assert_allclose(got.view(-1).detach().cpu().numpy(), z["predict"], rtol=RTOL, atol=ATOL)
    # the registered autograd: d sum(w * score) / d tables equals the ctypes path's gradients
    head, rel, tail = _tables(m, "kg")
    t2 = torch.stack([it["tail_id"], it["neg_tail_id"]], 1)
    w = torch.linspace(0.5, 1.5, t2.numel(), device=dev()).view_as(t2)
    m.zero_grad()
    s1 = ns.kge_score_grouped(mid, head, rel, tail, it["head_id"], it["relation_id"], 0, t2, m.margin, 0.0, False, 0)
    (s1 * w).sum().backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    s2 = ops.kge_score_grouped(model, tuple(head), tuple(rel), tuple(tail), it["head_id"], it["relation_id"], 0, t2, m.margin)
    (s2 * w).sum().backward()
    assert torch.equal(s1, s2)
    for k, p in m.named_parameters():
        if p.grad is None:
            assert k not in g1
            continue
        ref = p.grad.cpu().numpy()
        np.testing.assert_allclose(g1[k].cpu().numpy(), ref, rtol=2e-4, atol=2e-6 * (np.abs(ref).max() + 1e-12) + 1e-9, err_msg=k)
    # full sort + fused masked top-k
    users = torch.from_numpy(z["fs.users"]).to(dev())
    head, rel, tail = _tables(m, "rec")
    fs = ns.kge_full_sort(mid, head, rel, tail, users, None, m._rec_rel_full_sort(), 0, nI, m.margin)
    np.testing.assert_allclose(fs.detach().cpu().numpy(), z["fs.scores"], rtol=RTOL, atol=ATOL)
    hr, hc = torch.from_numpy(z["eval.history_rows"]).to(dev()), torch.from_numpy(z["eval.history_cols"]).to(dev())
    with torch.no_grad():
        v1, i1 = ns.kge_full_sort_topk(mid, head, rel, tail, users, None, m._rec_rel_full_sort(), 0, nI, 10, hr, hc, True,
                                       m.margin)
        m.gemm_full_sort = False  # the model's gather-kernel path: the registered op wraps the same kernel (bilinear scorers
        v2, i2 = m.full_sort_topk({"user_id": users}, 10, (hr, hc))  # otherwise evaluate through the GEMM: test_gpu_fullsort_gemm.py)
    assert torch.equal(v1, v2) and torch.equal(i1, i2)


@pytest.mark.parametrize("model", ["TransE", "RotatE", "ComplEx", "DistMult", "RESCAL", "HolE", "Analogy"])
def test_calculate_loss_through_registered_ops_equals_ctypes_path(ns, model, monkeypatch):
    dim = 16 if model == "RESCAL" else 64
    nU, nI, nE, nR = 60, 40, 120, 6
    tables = random_tables(model, nU, nE, nR, dim, seed=2)
    rng = np.random.default_rng(3)
    K = 4
    P = 24
    users, items = rng.integers(1, nU, P), rng.integers(1, nI, P)
    inter = to_dev(dict(user_id=np.tile(users, K), item_id=np.tile(items, K), neg_item_id=rng.integers(1, nI, P * K),
                        head_id=rng.integers(1, nE, 80), relation_id=rng.integers(1, nR - 1, 80),
                        tail_id=rng.integers(1, nE, 80), neg_tail_id=rng.integers(1, nE, 80)))
    neg = {"distribution": "uniform", "sample_num": K, "alpha": 1.0, "dynamic": False, "candidate_num": 0}
    out = {}
    for mode in ("torch_ops", "ctypes"):
        monkeypatch.setenv("HW_CTYPES_ONLY", "1" if mode == "ctypes" else "0")
        m = build(model, tables, dim, nU, nI, nE, nR, train_neg_sample_args=neg)
        loss = m.calculate_loss(inter)
        loss = sum(loss) if isinstance(loss, tuple) else loss
        loss.backward()
        out[mode] = (loss.item(), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()})
    assert out["torch_ops"][0] == out["ctypes"][0]
    for k, ref in out["ctypes"][1].items():
        np.testing.assert_allclose(out["torch_ops"][1][k], ref, rtol=2e-4, atol=2e-6 * (np.abs(ref).max() + 1e-12) + 1e-9, err_msg=k)


def test_fused_adam_sinks_through_registered_ops(ns):
    """Three steps with FusedAdam (gradients accumulated into its sinks by pair_loss_backward) == torch Adam."""
    import copy

    import hopwise_amd.model as M
    from hopwise_amd.optim import FusedAdam

    nU, nI, nE, nR, d = 60, 40, 120, 6, 32
    torch.manual_seed(1)
    m1 = M.RotatE(make_config(dev(), d, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev())
    m2 = copy.deepcopy(m1)
    o1, o2 = torch.optim.Adam(m1.parameters(), lr=1e-2), FusedAdam(m2.parameters(), lr=1e-2)
    rng = np.random.default_rng(2)
    for _ in range(3):
        it = {k: torch.from_numpy(rng.integers(1, hi, 96)).to(dev()) for k, hi in (
            ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1),
            ("tail_id", nE), ("neg_tail_id", nE))}
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m.calculate_loss(it).backward()
            o.step()
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        np.testing.assert_allclose(p2.detach().cpu().numpy(), p1.detach().cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_registered_sampler_equals_ctypes_sampler(ns):
    from hopwise_amd import ops

    rng = np.random.default_rng(0)
    n_keys, value_num = 50, 400
    used = [np.unique(rng.integers(1, value_num, rng.integers(0, 60))) for _ in range(n_keys)]
    ptr = np.zeros(n_keys + 1, np.int64)
    ptr[1:] = np.cumsum([len(u) for u in used])
    idx = np.concatenate(used).astype(np.int64)
    up, ui = torch.from_numpy(ptr).to(dev()), torch.from_numpy(idx).to(dev())
    keys = torch.from_numpy(rng.integers(0, n_keys, 333)).to(dev())
    a = ns.kg_neg_sample(up, ui, keys, 7, value_num, 2020, 5, None, None, None)
    b = ops.neg_sample(up, ui, keys, 7, value_num, 2020, 5)
    assert torch.equal(a, b)
    got = a.view(7, -1).cpu().numpy()
    for j in range(7):
        for i, k in enumerate(keys.cpu().numpy()):
            assert 1 <= got[j, i] < value_num and got[j, i] not in used[k]


def test_registered_pgpr_expand_and_collect_equal_ctypes(ns):
    from hopwise_amd import pgpr_ops as P
    from tests.test_gpu_pgpr import build_model

    g, z, orc, m = build_model("default")
    users = torch.from_numpy(z["users"]).to(dev())
    hops = [5, 3, 2]
    res = m.beam_search_device(users, hops, keep_trace=True, keep_buffers=True)
    buf, gr = res.buffers, m.graph
    meta = [gr.n_users, gr.n_entities, gr.n_items, gr.ui_relation, gr.max_degree]
    # replay hop 0's expansion from its trace: rows = users, top-k from the trace
    tr = res.trace[0]
    n0 = len(users)
    paths0, probs0 = buf["paths"][0][:n0].contiguous(), buf["probs"][0][:n0].contiguous()
    pout, qout, nout = ns.pgpr_expand(gr.ptr, gr.rel, gr.dst, meta, paths0, probs0, tr["acts"].contiguous(),
                                      tr["n_acts"].contiguous(), res.n_rows_per_hop[0:1].contiguous(),
                                      tr["tk_idx"].contiguous(), tr["tk_val"].contiguous(), n0 * hops[0])
    n1 = int(res.n_rows_per_hop[1].item())
    assert int(nout.item()) == n1
    assert torch.equal(pout[:n1], buf["paths"][1][:n1]) and torch.equal(qout[:n1], buf["probs"][1][:n1])
    # collect_scores on the leaves
    U64, E64, R64 = m._collect_tables()
    outs = ns.pgpr_collect(gr.ptr, gr.rel, gr.dst, meta, U64, E64, R64, users, m._pos_bits, res.paths, res.probs, res._n,
                           len(hops) + 1, max(m.topk))
    scores, sel = m.collect_scores_device(users, res)
    assert torch.equal(outs[0], scores)
    for name, t in zip(("item", "score", "leaf", "count", "row"), outs[1:]):
        cnt = sel["count"].cpu().numpy()
        a, b = t.cpu().numpy(), sel[name].cpu().numpy()
        if a.ndim == 2:
            for r in range(len(cnt)):
                assert np.array_equal(a[r, : cnt[r]], b[r, : cnt[r]]), name
        else:
            assert np.array_equal(a, b), name


def test_cpu_tensors_are_refused(ns):
    with pytest.raises(RuntimeError):
        ns.kge_full_sort(0, [torch.zeros(4, 8)], [torch.zeros(2, 8)], [torch.zeros(4, 8)], torch.zeros(2, dtype=torch.int64),
                         None, 0, 0, 4, 0.0)
