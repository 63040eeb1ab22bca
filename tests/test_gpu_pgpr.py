"""PGPR device path vs the reference's golden run and vs the oracle (build tie convention)."""

import os

import numpy as np
import pytest
import torch

from oracle import pgpr as op
from tests.helpers import FakeKGDataset, pgpr_config
from tests.test_oracle_golden_pgpr import TAGS, beam_names, dec_paths, load_oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def dev():
    return torch.device("cuda", 0)


def build_model(tag, dropout="identity"):
    from hopwise_amd.model.knowledge_aware_recommender import PGPR

    g, z, orc = load_oracle(tag, tie_rule="stable")
    ds = FakeKGDataset(int(g["n_users"]), int(g["n_items"]), int(g["n_entities"]), int(g["n_relations"]),
                       int(g["ui_relation"]), g["inter_users"], g["inter_items"], g["heads"], g["rels"], g["tails"],
                       g["positives"], g["user_emb"], g["entity_emb"], g["relation_emb"])
    cfg = pgpr_config(dev(), int(z["max_acts"]), int(z["state_history"]), (25, 5, 1), bool(z["fix_scores_sorting_bug"]),
                      tuple(int(x) for x in z["topk"]), dropout)
    m = PGPR(cfg, ds).to(dev())
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")})
    return g, z, orc, m


def paths_to_records(paths):
    rec = np.zeros((len(paths), 12), np.int32)
    for i, p in enumerate(paths):
        rec[i, 0], rec[i, 1] = p[0][2], len(p)
        for j, (r, t, n) in enumerate(p[1:]):
            rec[i, 2 + 3 * j : 5 + 3 * j] = (r, t, n)
    return rec


def prim_paths(z):
    flat, off = z["prim.paths_flat"], z["prim.paths_off"]
    return [[(int(r), int(t), int(n)) for r, t, n in flat[off[i] : off[i + 1]]] for i in range(len(off) - 1)]


@pytest.mark.parametrize("tag", TAGS)
def test_actions_and_state_kernels(tag):
    from hopwise_amd import pgpr_ops as P

    g, z, orc, m = build_model(tag)
    paths = prim_paths(z)
    n = len(paths)
    recs = paths_to_records(paths)
    roots = sorted({p[0][2] for p in paths})
    recs[:, 11] = [roots.index(p[0][2]) for p in paths]  # batch index of the root user
    rec = torch.from_numpy(recs).to(dev())
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    acts = torch.full((n, m.act_dim, 2), -7, dtype=torch.int32, device=dev())
    n_acts = torch.zeros(n, dtype=torch.int32, device=dev())
    su, se = P.user_scores(m.graph, *m._t64, torch.tensor(roots, device=dev()))
    P.actions(m.graph, su, se, m._score_rel_entity, rec, n_rows, n, m.max_acts, m.fix_scores_sorting_bug, acts, n_acts, None)
    st = torch.empty((n, m.state_dim), dtype=torch.float32, device=dev())
    P.state(m.graph, *m._t32, m.state_history, rec, n_rows, n, st)
    acts, n_acts, st = acts.cpu().numpy(), n_acts.cpu().numpy(), st.cpu().numpy()
    aflat, aoff = z["prim.acts_flat"], z["prim.acts_off"]
    n_trim = 0
    for i, path in enumerate(paths):
        want = orc.get_actions(path)  # build convention (stable ties)
        got = [tuple(int(x) for x in a) for a in acts[i, : n_acts[i]]]
        assert got == want, (tag, i)
        ref = [(int(r), int(nn)) for r, nn in aflat[aoff[i] : aoff[i + 1]]]
        if got != ref:  # vs the reference: only float64 score ties may differ
            cand, sc = orc.action_scores(path)
            diff = set(got) ^ set(ref)
            kth = np.sort(sc)[-orc.max_acts]
            assert len(got) == len(ref) and all(sc[cand.index(a)] == kth for a in diff) or m.fix_scores_sorting_bug
        n_trim += len(ref) == orc.max_acts + 1
    if tag.startswith("trim"):
        assert n_trim > 0
    np.testing.assert_array_equal(st, z["prim.states"].astype(np.float32))


@pytest.mark.parametrize("queue", [True, False])
def test_actions_trim_on_high_degree_rows(queue, monkeypatch):
    """Rows of every degree class of hw_pgpr_actions -- wave-resident trim (8 / 16 / 32 keys per lane), block-resident
    trim (16 / 32 keys per thread) and the radix-select fallback past 8 192 neighbours -- against the oracle's
    get_actions, with duplicate edges so that equal float64 scores meet the threshold; with and without the work queues."""
    from hopwise_amd import pgpr_ops as P
    from hopwise_amd.model.knowledge_aware_recommender import PGPR
    from oracle.pgpr import CKG, ENTITY, SELF_LOOP, USER, PGPROracle

    monkeypatch.setattr(P, "QUEUE_TRIM_ROWS", queue)
    rng = np.random.default_rng(12)
    n_users, n_items, n_ent, n_rel, d = 9500, 12, 40, 4, 8
    ui = n_rel - 1
    degs = {1: 9100, 2: 5000, 3: 3000, 4: 1500, 5: 700, 6: 300, 7: 251, 8: 90}
    iu, ii = [], []
    for item, k in degs.items():
        us = rng.choice(np.arange(1, n_users), size=k, replace=False)
        us[: k // 50] = us[k // 50 : 2 * (k // 50)]  # some users interact twice: duplicate edges, equal scores
        iu.append(us), ii.append(np.full(k, item))
    iu, ii = np.concatenate(iu), np.concatenate(ii)
    heads = rng.integers(1, n_ent, 60)
    rels = rng.integers(1, ui, 60)
    tails = rng.integers(1, n_ent, 60)
    U = rng.standard_normal((n_users, d))
    E = rng.standard_normal((n_ent, d))
    R = rng.standard_normal((n_rel, d))
    positives = np.zeros((n_users, 1), np.int64)
    ds = FakeKGDataset(n_users, n_items, n_ent, n_rel, ui, iu, ii, heads, rels, tails, positives, U, E, R)
    m = PGPR(pgpr_config(dev(), 250, 1, (25, 5, 1), False, (10,), "identity"), ds).to(dev())
    g = CKG(n_users, n_ent, ui, iu, ii, heads, rels, tails)
    orc = PGPROracle(g, n_items, U, E, R, positives, {}, max_acts=250, tie_rule="stable")
    u0 = int(iu[0])
    paths = [[(SELF_LOOP, USER, u0), (ui, ENTITY, item)] for item in degs] + [[(SELF_LOOP, USER, u0)]]
    n = len(paths)
    recs = paths_to_records(paths)
    rec = torch.from_numpy(recs).to(dev())  # batch index 0: one root user
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    acts = torch.full((n, m.act_dim, 2), -7, dtype=torch.int32, device=dev())
    n_acts = torch.zeros(n, dtype=torch.int32, device=dev())
    su, se = P.user_scores(m.graph, *m._t64, torch.tensor([u0], device=dev()))
    scratch = torch.empty((n, int(m.graph.max_degree)), dtype=torch.int64, device=dev())
    P.actions(m.graph, su, se, m._score_rel_entity, rec, n_rows, n, m.max_acts, False, acts, n_acts, scratch)
    acts, n_acts = acts.cpu().numpy(), n_acts.cpu().numpy()
    for i, path in enumerate(paths):
        want = orc.get_actions(path)
        got = [tuple(int(x) for x in a) for a in acts[i, : n_acts[i]]]
        if got != want:  # the score tables add two dot products where the oracle dots a summed vector: last-ulp ties only
            cand, sc = orc.action_scores(path)
            kth = np.sort(sc)[-orc.max_acts]
            diff = set(got) ^ set(want)
            assert len(got) == len(want) and all(abs(sc[cand.index(a)] - kth) <= 1e-12 * abs(kth) for a in diff), i
        assert len(got) == min(251, 1 + len(orc._candidates(path, "sorted")))


@pytest.mark.parametrize("tag", ["default", "trim20"])
def test_policy_forward_kernels(tag):
    from hopwise_amd import pgpr_ops as P

    g, z, orc, m = build_model(tag)
    st = torch.from_numpy(z["prim.states"].astype(np.float32)).to(dev())
    n = st.shape[0]
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    n_acts = torch.from_numpy(z["prim.masks"].sum(1).astype(np.int32)).to(dev())
    for d1, d2, want in ((None, None, z["prim.fwd_probs"]), (z["prim.drop1"], z["prim.drop2"], z["prim.fwd_probs_drop"])):
        buf = dict(state=st, x1=torch.empty(n, 512, device=dev()), x2=torch.empty(n, 256, device=dev()),
                   logits=torch.empty(n, m.act_dim, device=dev()))
        drops = None if d1 is None else [(torch.from_numpy(d1).to(dev()), torch.from_numpy(d2).to(dev()))]
        m._policy_logits(buf, n_rows, n, drops, 0)
        for k in sorted({min(5, m.act_dim), min(21, m.act_dim), min(64, m.act_dim), m.act_dim}):  # arg-max and sort kernels
            idx = torch.empty((n, k), dtype=torch.int32, device=dev())
            val = torch.empty((n, k), dtype=torch.float32, device=dev())
            probs = torch.empty((n, m.act_dim), dtype=torch.float32, device=dev())
            P.policy_topk(buf["logits"], m.act_dim, n_acts, n_rows, n, k, idx, val, probs)
            np.testing.assert_allclose(probs.cpu().numpy(), want, rtol=1e-4, atol=1e-7)
            # top-k of the kernel == top-k of its own probs under (value desc, index asc), on p + 1
            pp = probs.cpu().numpy()
            na = n_acts.cpu().numpy()
            key = np.where(np.arange(m.act_dim)[None, :] < na[:, None], pp + np.float32(1.0), np.float32(0.0))
            order = np.lexsort((np.broadcast_to(np.arange(m.act_dim), key.shape), -key.astype(np.float64)), axis=1)[:, :k]
            gi = idx.cpu().numpy()
            for r in range(n):
                kk = min(k, na[r])
                assert list(gi[r, :kk]) == list(order[r, :kk]) and all(gi[r, kk:] == -1), (k, r)
                np.testing.assert_array_equal(val.cpu().numpy()[r, :kk], key[r, order[r, :kk]])


safe_leaves = op.safe_leaves


# Overlap of the device beams with the oracle's / the reference's, as (leaves, safe, device ∩ oracle, device ∩ reference)
# per (tag, hops) measured on MI355X (deterministic: identity dropout, deterministic split-K).  A lower number is a
# regression and fails; the differences that remain are float32 ties at a top-k cut, each one proven below.
BEAM_PINNED = {}
_pin_path = os.path.join(os.path.dirname(__file__), "golden", "pgpr_beam_overlap.json")
if os.path.exists(_pin_path):
    import json as _json

    BEAM_PINNED = _json.load(open(_pin_path))


@pytest.mark.parametrize("tag", TAGS)
def test_beam_search_device(tag):
    g, z, orc, m = build_model(tag)
    users = torch.from_numpy(z["users"]).to(dev())
    observed = {}
    for hn in beam_names(z):
        hops = [int(x) for x in z[f"beam.{hn}.hops"]]
        res = m.beam_search_device(users, hops)
        gp, gq = res.numpy()
        got_paths, got_probs = m.decode_paths(gp, gq)
        got = {tuple((-1 if r == "self_loop" else r, 0 if t == "user" else 1, n) for r, t, n in p): np.asarray(q)
               for p, q in zip(got_paths, got_probs)}
        opaths, oprobs, safe = safe_leaves(orc, z["users"], hops)
        okeys = [tuple(map(tuple, p)) for p in opaths]
        unsafe = {k for k, s in zip(okeys, safe) if not s}
        n_safe = len(okeys) - len(unsafe)
        # 1. the proof: an oracle leaf the device did not emit has a hop whose selected value sits within 2e-6 of the
        #    first value that was cut (float32 rounding can flip exactly those selections and no others)
        missing = [k for k in okeys if k not in got]
        assert all(k in unsafe for k in missing), (tag, hn, [k for k in missing if k not in unsafe][:3])
        # 2. every leaf both sides emit carries the same hop probabilities
        for k, q in zip(okeys, oprobs):
            if k in got:
                np.testing.assert_allclose(got[k], np.asarray(q, np.float32), rtol=3e-6, atol=0)
        overlap = len(okeys) - len(missing)
        # 3. the reference's own beams: identical to the device's except at those same ties
        ref = {tuple(map(tuple, p)) for p in dec_paths(z[f"beam.{hn}.paths"])}
        ref_overlap = len(ref & set(got))
        if not m.fix_scores_sorting_bug:  # (under the flag the reference's argsort tie order is unpinnable)
            stray = [k for k in ref if k not in got and k not in unsafe]
            assert not stray, (tag, hn, "reference leaves missing without a tie at a cut", stray[:3])
        observed[hn] = [len(okeys), n_safe, overlap, ref_overlap, len(ref)]
        msg = (f"{tag}/{hn}: {len(okeys)} oracle leaves, {n_safe} safe ({n_safe / max(len(okeys), 1):.4f}), device∩oracle "
               f"{overlap} ({overlap / max(len(okeys), 1):.4f}), device∩reference {ref_overlap}/{len(ref)}")
        pin = BEAM_PINNED.get(f"{tag}/{hn}")
        if pin is not None:
            assert overlap >= pin[2] and (m.fix_scores_sorting_bug or ref_overlap >= pin[3]), ("regression: " + msg, pin)
        assert overlap >= 0.97 * len(okeys), msg
        # children are emitted in row order: users appear in batch order
        first_user = [int(r[0]) for r in gp]
        order = [u for i, u in enumerate(first_user) if i == 0 or first_user[i - 1] != u]
        assert order == [int(u) for u in z["users"] if int(u) in set(first_user)]
    dump = os.environ.get("HW_DUMP_BEAM_OVERLAP")
    if dump:
        import json

        cur = json.load(open(dump)) if os.path.exists(dump) else {}
        cur.update({f"{tag}/{hn}": v for hn, v in observed.items()})
        json.dump(cur, open(dump, "w"), indent=1, sort_keys=True)


def test_reference_dropout_mode_runs_and_is_stochastic():
    g, z, orc, m = build_model("default", dropout="reference")
    users = torch.from_numpy(z["users"]).to(dev())
    a = m.beam_search_device(users, [4, 3, 2]).numpy()
    b = m.beam_search_device(users, [4, 3, 2]).numpy()
    assert a[0].shape[1] == 12 and len(a[0]) > 0
    assert a[0].shape != b[0].shape or not np.array_equal(a[0], b[0]) or not np.array_equal(a[1], b[1])


def test_mlp_layer_vs_fp64():
    """The fp32-MFMA layer against a float64 product on ragged shapes (K, N not multiples of the tile)."""
    from hopwise_amd import pgpr_ops as P

    rng = np.random.default_rng(0)
    for M, K, N in ((1, 16, 251), (130, 400, 512), (257, 24, 21), (64, 512, 256), (1000, 100, 71)):
        a = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        at, wt, bt = (torch.from_numpy(x).to(dev()) for x in (a, w, b))
        out = torch.empty((M, N), device=dev())
        P.mlp_layer(at, wt, bt, out, None, M, True)
        ref = a.astype(np.float64) @ w.astype(np.float64).T + b
        ref = np.where(ref > 0, ref, np.expm1(ref))
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
        out2 = torch.empty((M, N), device=dev())
        P.mlp_layer(at, wt, bt, out2, torch.tensor([M // 2], dtype=torch.int32, device=dev()), M, False)
        ref2 = a.astype(np.float64) @ w.astype(np.float64).T + b
        np.testing.assert_allclose(out2.cpu().numpy()[: M // 2], ref2[: M // 2], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("tag", TAGS)
def test_collect_scores_device_on_reference_beams(tag):
    """Device collect_scores fed with the REFERENCE's beams must reproduce the reference's score matrix and
    explanation list bit for bit (float64 scores -> float32), quirks included."""
    from hopwise_amd.model.knowledge_aware_recommender import BeamResult

    g, z, orc, m = build_model(tag)
    users_np = z["users"]
    users = torch.from_numpy(users_np).to(dev())
    first = {int(u): i for i, u in reversed(list(enumerate(users_np)))}
    for hn in beam_names(z):
        ref_paths = dec_paths(z[f"beam.{hn}.paths"])
        rec = paths_to_records(ref_paths)
        rec[:, 11] = [first[int(p[0][2])] for p in ref_paths]
        probs = np.zeros((len(ref_paths), 3), np.float32)
        probs[:, : z[f"beam.{hn}.probs"].shape[1]] = z[f"beam.{hn}.probs"]
        res = BeamResult(torch.from_numpy(rec).to(dev()), torch.from_numpy(probs).to(dev()),
                         torch.tensor([len(rec)], dtype=torch.int32, device=dev()))
        scores, sel = m.collect_scores_device(users, res, n_nodes=4)
        np.testing.assert_array_equal(scores.cpu().numpy(), z[f"collect.{hn}.scores"])
        expl = m.explanations(users, res, sel)
        assert [[e[0], e[1]] for e in expl] == z[f"collect.{hn}.expl_uis"].tolist()
        np.testing.assert_allclose([e[2] for e in expl], z[f"collect.{hn}.expl_score"], rtol=1e-14)
        want_paths = dec_paths(z[f"collect.{hn}.expl_paths"])
        got_paths = [[(-1 if r == "self_loop" else r, 0 if t == "user" else 1, n) for r, t, n in e[3]] for e in expl]
        assert got_paths == want_paths


def test_explain_end_to_end():
    g, z, orc, m = build_model("default")
    users = torch.from_numpy(z["users"]).to(dev())
    scores, expl = m.explain({"user_id": users})
    assert tuple(scores.shape) == (len(z["users"]), int(g["n_items"]))
    fs = m.full_sort_predict({"user_id": users})
    np.testing.assert_array_equal(fs.cpu().numpy(), scores.cpu().numpy())
    finite = torch.isfinite(scores).sum(1).cpu().numpy()
    assert finite.max() <= max(m.topk) and len(expl) == int(finite.sum()) or len(expl) >= int(finite.sum())
    for u, i, s, path in expl:
        assert path[0] == ("self_loop", "user", u) and len(path) == 4


def test_gemm_all_layouts_vs_fp64():
    from hopwise_amd import ops

    rng = np.random.default_rng(1)
    for M, K, N in ((1, 1, 1), (37, 200, 129), (300, 17, 260), (128, 128, 128), (1000, 403, 70), (130, 5003, 40)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K)).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        At, Bt = torch.from_numpy(A).to(dev()), torch.from_numpy(B).to(dev())
        AT, BT = At.t().contiguous(), Bt.t().contiguous()  # stored [K, M] and [K, N]
        tol = dict(rtol=2e-5, atol=2e-5 * np.sqrt(K))
        np.testing.assert_allclose(ops.gemm(At, Bt, True, True).cpu().numpy(), ref, **tol)
        np.testing.assert_allclose(ops.gemm(At, BT, True, False).cpu().numpy(), ref, **tol)
        np.testing.assert_allclose(ops.gemm(AT, BT, False, False).cpu().numpy(), ref, **tol)
        np.testing.assert_allclose(ops.gemm(AT, Bt, False, True).cpu().numpy(), ref, **tol)
    # autograd wrapper
    x = torch.randn(50, 30, device=dev(), requires_grad=True)
    w = torch.randn(20, 30, device=dev(), requires_grad=True)
    y = ops.matmul_nt(x, w)
    g = torch.randn_like(y)
    y.backward(g)
    np.testing.assert_allclose(x.grad.cpu().numpy(), (g.double() @ w.detach().double()).cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(w.grad.cpu().numpy(), (g.double().t() @ x.detach().double()).cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_gemm_pipelined_loop_on_padded_operands():
    """Operands with 16-byte aligned rows take the software-pipelined loop, whose fetch clamps addresses instead of
    testing edges: inner extents that are not multiples of 4 (views into padded buffers), K below / not a multiple of
    the 16-deep tile, ragged M and N, and the split-K form (long K, small output), all four layouts, against float64."""
    from hopwise_amd import ops

    rng = np.random.default_rng(5)

    def padded(x):  # [r, c] -> view [r, c] of a buffer with ld = c rounded up to 4 (+4 when already a multiple)
        r, c = x.shape
        ld = (c + 3) // 4 * 4 + (4 if c % 4 == 0 else 0)
        buf = torch.full((r, ld), float("nan"), device=dev())
        buf[:, :c] = torch.from_numpy(x).to(dev())
        return buf[:, :c]

    for M, K, N in ((5, 3, 7), (37, 201, 129), (300, 18, 259), (129, 130, 131), (1000, 403, 70), (130, 5003, 41),
                    (2048, 6001, 200), (4, 4, 4),
                    # 192 < N <= 208: the whole-row 16x16x4 kernel (gemm_f32_n208_kernel), ragged M / K, split-K
                    (5, 3, 200), (129, 130, 208), (1000, 403, 193), (300, 20011, 205), (144, 2048, 200), (4, 4, 196)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((N, K)).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        At, Bt = padded(A), padded(B)
        AT, BT = padded(np.ascontiguousarray(A.T)), padded(np.ascontiguousarray(B.T))
        tol = dict(rtol=2e-5, atol=2e-5 * np.sqrt(K))
        for a, b, akc, bkc in ((At, Bt, True, True), (At, BT, True, False), (AT, BT, False, False), (AT, Bt, False, True)):
            out = padded(np.zeros((M, N), np.float32))
            ops.gemm(a, b, akc, bkc, out=out)
            np.testing.assert_allclose(out.cpu().numpy(), ref, err_msg=f"{M}x{K}x{N} {akc} {bkc}", **tol)


@pytest.mark.parametrize("N", [256, 252, 251, 40])
def test_philox_dropout_masks_of_the_policy_layers(N):
    """The keep-1/2 masks drawn inside the GEMM epilogue (quad-shared Philox blocks when N % 4 == 0, one evaluation per
    element otherwise), the split-K reduction and the first-layer lookup follow hw_common.h:philox_bits word for word."""
    from hopwise_amd import pgpr_ops as P
    from oracle.philox import philox_words

    rng = np.random.default_rng(N)
    seed, offset = 0x1234567890ABCDEF, 77
    for M, K in ((300, 64), (130, 512)):  # the second shape takes the split-K path
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev())
        w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(dev())
        b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev())
        plain = torch.empty((M, N), device=dev())
        drop = torch.empty((M, N), device=dev())
        P.mlp_layer(a, w, b, plain, None, M, True)
        P.mlp_layer(a, w, b, drop, None, M, True, None, 2, seed, offset)
        keep = (philox_words(seed, offset, np.arange(M * N, dtype=np.uint64)) & 1).reshape(M, N).astype(np.float32)
        np.testing.assert_array_equal(drop.cpu().numpy(), plain.cpu().numpy() * (2.0 * keep))


@pytest.mark.parametrize("k", [9, 33, 64, 100])  # <= 64: threshold selection + 64-key sort; above: the 256-key sort
def test_policy_topk_sort_edge_cases(k):
    """Saturated rows (one probability rounds p + 1 to 2.0), equal logits (index-ascending ties), single action."""
    from hopwise_amd import pgpr_ops as P

    act_dim = 251
    logits = np.zeros((5, act_dim), np.float32)
    logits[0, 7] = 40.0  # p[7] = 1 - tiny -> p + 1 == 2.0, everything else collapses onto 1.0
    logits[1, :] = 0.5  # all equal
    logits[2, :] = np.linspace(-3, 3, act_dim)
    logits[3, 0] = 1.0
    logits[4, :] = np.random.default_rng(0).standard_normal(act_dim)
    na = np.array([100, 251, 251, 1, 37], np.int32)
    lt = torch.from_numpy(logits).to(dev())
    idx = torch.empty((5, k), dtype=torch.int32, device=dev())
    val = torch.empty((5, k), dtype=torch.float32, device=dev())
    probs = torch.empty((5, act_dim), dtype=torch.float32, device=dev())
    P.policy_topk(lt, act_dim, torch.from_numpy(na).to(dev()), torch.tensor([5], dtype=torch.int32, device=dev()), 5, k,
                  idx, val, probs)
    pp, gi, gv = probs.cpu().numpy(), idx.cpu().numpy(), val.cpu().numpy()
    key = np.where(np.arange(act_dim)[None, :] < na[:, None], pp + np.float32(1.0), np.float32(0.0))
    order = np.lexsort((np.broadcast_to(np.arange(act_dim), key.shape), -key.astype(np.float64)), axis=1)[:, :k]
    for r in range(5):
        kk = min(k, na[r])
        assert list(gi[r, :kk]) == list(order[r, :kk]) and all(gi[r, kk:] == -1), r
        np.testing.assert_array_equal(gv[r, :kk], key[r, order[r, :kk]])
    assert gv[0, 0] == 2.0 and gi[0, 0] == 7 and gv[3, 0] == 2.0


@pytest.mark.parametrize("state_history,hidden", [(1, 512), (2, 256), (1, 640)])
def test_layer1_lookup_runs_of_siblings_equal_row_by_row(state_history, hidden):
    """hw_pgpr_layer1_lookup on >= 65 536 rows lets a wave walk 16 consecutive rows and keep the user / last-node rows of
    sibling paths in registers; the result must equal the row-per-wave launches (< 65 536 rows each) bit for bit, with
    Philox dropout on.  Paths: runs of 1..64 siblings, lengths 1..3, both node types, self-loop relations."""
    from hopwise_amd import pgpr_ops as P

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(hidden + state_history)
    nU, nE, nR, ROW = 300, 500, 9, 12
    n = 70001
    paths = np.zeros((n, ROW), np.int32)
    i = 0
    while i < n:
        run = int(rng.integers(1, 65))
        uid, length = int(rng.integers(1, nU)), int(rng.integers(1, 4))
        parent = [(int(rng.integers(-1, nR)), int(rng.integers(0, 2)), int(rng.integers(1, nU))) for _ in range(2)]
        for j in range(min(run, n - i)):
            p = paths[i + j]
            p[0], p[1] = uid, length
            hops = parent[: max(length - 2, 0)] + [(int(rng.integers(-1, nR)), int(rng.integers(0, 2)), int(rng.integers(1, nU)))]
            for h, (r, t, node) in enumerate(hops[: length - 1] if length > 1 else []):
                p[2 + 3 * h: 5 + 3 * h] = (r, t, node)
            # last hop differs between siblings, earlier hops are the parent's
        i += run
    g = torch.Generator(device=dev).manual_seed(1)
    tabs = [torch.randn((rows, hidden), device=dev, generator=g) for rows in
            ([nU, nU, nE] + [nU, nE, nR] * state_history)]
    bias = torch.randn(hidden, device=dev, generator=g)
    pt = torch.from_numpy(paths).to(dev)
    whole = torch.empty((n, hidden), device=dev)
    P.layer1_lookup(tabs, bias, state_history, pt, torch.tensor([n], dtype=torch.int32, device=dev), n, whole,
                    drop_mode=2, seed=3, offset=9)
    # row-per-wave reference: the same launch with max_rows below the threshold sees the same rows only if the Philox index
    # (row * H + c) is the same, so run it on the full path buffer with a row count that stops early, slice by slice
    ref = torch.empty((n, hidden), device=dev)
    P.layer1_lookup(tabs, bias, state_history, pt, torch.tensor([n], dtype=torch.int32, device=dev), 60000, ref,
                    drop_mode=2, seed=3, offset=9)  # max_rows only sizes the grid / picks the run length
    assert torch.equal(whole, ref)
    assert float(whole.abs().sum()) > 0
