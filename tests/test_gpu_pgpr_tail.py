"""The tail of a beam-search hop as one launch (hw_pgpr_topk_expand: masked softmax + top-k + expand) and the compact
16-byte leaves of the last hop, against the two-launch form with full records (policy_topk + expand, each pinned by the
reference goldens in tests/test_gpu_pgpr.py): every array bit for bit."""

import numpy as np
import pytest
import torch

from tests.test_gpu_pgpr import TAGS, beam_names, build_model, dev

pytestmark = pytest.mark.gpu


def _search(m, users, hops, **kw):
    m._drop_calls = 0
    res = m.beam_search_device(users, hops, **kw)
    return res, res.numpy()


@pytest.mark.parametrize("tag", TAGS)
def test_fused_tail_and_compact_leaves_equal_the_two_launch_form(tag):
    g, z, orc, m = build_model(tag)
    users = torch.from_numpy(z["users"]).to(dev())
    hop_sets = [[int(x) for x in z[f"beam.{hn}.hops"]] for hn in beam_names(z)] + [[min(64, m.act_dim), 3, 2], [9, 1], [7]]
    for hops in hop_sets:
        _, (p0, q0) = _search(m, users, hops, fused_tail=False)
        _, (p1, q1) = _search(m, users, hops, fused_tail=True, compact_leaves=False)
        res2, (p2, q2) = _search(m, users, hops, fused_tail=True, compact_leaves=True)
        assert res2.leaves is not None and len(p0) > 0
        for p, q in ((p1, q1), (p2, q2)):
            np.testing.assert_array_equal(p, p0, err_msg=f"{tag} {hops}")
            if min(hops) > 8:  # the two-launch form ranks with the same kernel body (policy_topk_sort_kernel<true>)
                np.testing.assert_array_equal(q, q0, err_msg=f"{tag} {hops}")
            else:  # k <= 8 goes through policy_topk_kernel there (expf and a division per element): an ulp apart
                np.testing.assert_allclose(q, q0, rtol=3e-7, atol=0, err_msg=f"{tag} {hops}")
        np.testing.assert_array_equal(q2, q1)  # compact leaves against full records of the same launch: always bit for bit
        # action lists as (relation, dst) pairs instead of 2-byte edge offsets: the same children
        m.acts_u16 = False
        _, (p3, q3) = _search(m, users, hops, fused_tail=True, compact_leaves=True)
        m.acts_u16 = True
        np.testing.assert_array_equal(p3, p2)
        np.testing.assert_array_equal(q3, q2)
        # single rows through the selective rebuild, out-of-range entries as zero rows
        n = len(p0)
        idx = np.array([0, n - 1, n // 2, -1, n], np.int64)
        rp, rq = res2.rows(torch.from_numpy(idx))
        rp, rq = rp.cpu().numpy(), rq.cpu().numpy()
        np.testing.assert_array_equal(rp[:3], p0[idx[:3]])
        np.testing.assert_array_equal(rq[:3], q0[idx[:3]])
        assert not rp[3:].any() and not rq[3:].any()


@pytest.mark.parametrize("tag", ["default", "trim20", "hist2"])
def test_reference_dropout_stream_is_the_same_through_the_fused_tail(tag):
    """Philox policy dropout: the tail does not touch the stream (same _drop_calls, same children)."""
    g, z, orc, m = build_model(tag, dropout="reference")
    users = torch.from_numpy(z["users"]).to(dev())
    _, (p0, q0) = _search(m, users, [12, 10, 9], fused_tail=False)
    _, (p1, q1) = _search(m, users, [12, 10, 9], fused_tail=True, compact_leaves=True)
    np.testing.assert_array_equal(p1, p0)
    np.testing.assert_array_equal(q1, q0)


@pytest.mark.parametrize("tag", ["default", "trim70"])
def test_collect_on_compact_leaves_equals_collect_on_records(tag):
    g, z, orc, m = build_model(tag)
    users = torch.from_numpy(z["users"]).to(dev())
    for hops in ([25, 5, 1], [6, 4, 3], [10, 4], [12], [64, 64, 64]):  # the last one: enough leaves for the LDS mark pass
        full, _ = _search(m, users, hops, fused_tail=True, compact_leaves=False)
        comp, _ = _search(m, users, hops, fused_tail=True, compact_leaves=True)
        n_nodes = len(hops) + 1
        s0, sel0 = m.collect_scores_device(users, full, n_nodes=n_nodes)
        s1, sel1 = m.collect_scores_device(users, comp, n_nodes=n_nodes)
        assert comp._paths is None, "the collector must read the compact leaves in place"
        np.testing.assert_array_equal(s1.cpu().numpy(), s0.cpu().numpy())
        for k in ("item", "score", "leaf", "count", "row"):
            assert torch.equal(sel1[k], sel0[k]), (tag, hops, k)
        if len(hops) == 3:  # shorter walks rarely end on an unseen item of this small fixture
            assert int(sel0["count"].sum()) > 0
        assert m.explanations(users, comp, sel1) == m.explanations(users, full, sel0)


def test_output_capacity_is_respected():
    """Children past out_capacity are dropped, the row count is capped, nothing is written past the buffers -- as expand."""
    from hopwise_amd import pgpr_ops as P

    g, z, orc, m = build_model("default")
    users = torch.from_numpy(z["users"]).to(dev())
    res = m.beam_search_device(users, [6, 9], fused_tail=False, keep_buffers=True)
    buf = res.buffers
    n1 = int(buf["n_rows"][1].item())
    # redo the second hop's tail by hand at a capacity that cuts a row's children in the middle
    paths, probs, n_rows = buf["paths"][1], buf["probs"][1], buf["n_rows"][1:2]
    P.actions(m.graph, *P.user_scores(m.graph, *m._t64, users), m._score_rel_entity, paths, n_rows, paths.shape[0], m.max_acts,
              m.fix_scores_sorting_bug, buf["acts"], buf["n_acts"], buf["scratch"])
    m._drop_calls = 0
    m._policy_logits(buf, n_rows, paths.shape[0], None, 1, paths)
    full_n = int(torch.minimum(buf["n_acts"][:n1], torch.tensor(9, device=dev())).sum().item())
    cap = full_n - 3
    sentinel = -123
    outs = []
    for leaf_mode in (False, True):
        po = torch.full((cap + 8, 12), sentinel, dtype=torch.int32, device=dev())
        qo = torch.full((cap + 8, 3), -5.0, device=dev())
        lo = torch.full((cap + 8, 4), sentinel, dtype=torch.int32, device=dev())
        n_out = torch.zeros(1, dtype=torch.int32, device=dev())
        P.topk_expand(m.graph, paths, probs, buf["acts"], m.act_dim, buf["n_acts"], n_rows, paths.shape[0], buf["logits"], 9,
                      buf["offsets"], po, qo, lo if leaf_mode else None, n_out, cap)
        assert int(n_out.item()) == cap
        if leaf_mode:
            assert bool((lo[cap:] == sentinel).all()) and bool((po == sentinel).all())
            lp, lq = P.materialize_leaves(m.graph, paths, probs, lo, n_out, n_out=cap)
            outs.append((lp.cpu().numpy(), lq.cpu().numpy()))
        else:
            assert bool((po[cap:] == sentinel).all()) and bool((qo[cap:] == -5.0).all())
            outs.append((po[:cap].cpu().numpy(), qo[:cap].cpu().numpy()))
    # the two-launch form at the same capacity
    tk_i = torch.empty((paths.shape[0], 9), dtype=torch.int32, device=dev())
    tk_v = torch.empty((paths.shape[0], 9), device=dev())
    P.policy_topk(buf["logits"], m.act_dim, buf["n_acts"], n_rows, paths.shape[0], 9, tk_i, tk_v)
    po = torch.full((cap + 8, 12), sentinel, dtype=torch.int32, device=dev())
    qo = torch.full((cap + 8, 3), -5.0, device=dev())
    n_out = torch.zeros(1, dtype=torch.int32, device=dev())
    P.expand(m.graph, paths, probs, buf["acts"], m.act_dim, buf["n_acts"], n_rows, paths.shape[0], tk_i, tk_v, 9, buf["offsets"],
             po, qo, n_out, cap)
    assert int(n_out.item()) == cap
    for p, q in outs:
        np.testing.assert_array_equal(p, po[:cap].cpu().numpy())
        np.testing.assert_array_equal(q, qo[:cap].cpu().numpy())


def test_unsupported_shapes_fall_back():
    """k > 64 keeps the two launches (and full records); the search still runs and the result is a full BeamResult."""
    g, z, orc, m = build_model("default")
    users = torch.from_numpy(z["users"]).to(dev())[:2]
    res = m.beam_search_device(users, [3, 100])
    assert res.leaves is None and res.n > 0


@pytest.mark.parametrize("tag", [t for t in TAGS if "fix" not in t])
def test_u16_action_lists_decode_to_the_pair_lists(tag):
    """hw_pgpr_actions with flag bit 2: entry a of a row is the offset of action a's edge in the current node's CSR slice
    (0xFFFF = the self loop); decoded through the CSR it is the (relation, dst) pair of the int32 layout, for every row class
    (short rows, wave-trimmed, block-trimmed where the fixture has them)."""
    from hopwise_amd import pgpr_ops as P
    from tests.test_gpu_pgpr import paths_to_records, prim_paths

    g, z, orc, m = build_model(tag)
    paths = prim_paths(z)
    n = len(paths)
    recs = paths_to_records(paths)
    roots = sorted({p[0][2] for p in paths})
    recs[:, 11] = [roots.index(p[0][2]) for p in paths]
    rec = torch.from_numpy(recs).to(dev())
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    su, se = P.user_scores(m.graph, *m._t64, torch.tensor(roots, device=dev()))
    a32 = torch.full((n, m.act_dim, 2), -7, dtype=torch.int32, device=dev())
    n32 = torch.zeros(n, dtype=torch.int32, device=dev())
    P.actions(m.graph, su, se, m._score_rel_entity, rec, n_rows, n, m.max_acts, False, a32, n32, None)
    a16 = torch.full((n, P.act16_ld(m.act_dim)), -7, dtype=torch.int16, device=dev())
    n16 = torch.zeros(n, dtype=torch.int32, device=dev())
    P.actions(m.graph, su, se, m._score_rel_entity, rec, n_rows, n, m.max_acts, False, a16, n16, None, u16=True)
    assert torch.equal(n16, n32)
    ptr, rel, dst = (x.cpu().numpy() for x in (m.graph.ptr, m.graph.rel, m.graph.dst))
    a32, a16, cnt = a32.cpu().numpy(), a16.cpu().numpy().view(np.uint16), n32.cpu().numpy()
    n_trimmed = 0
    for i, p in enumerate(paths):
        t, nid = (0, p[0][2]) if len(p) == 1 else (p[-1][1], p[-1][2])
        lo = ptr[nid if t == 0 else int(g["n_users"]) + nid]
        assert a16[i, 0] == 0xFFFF and tuple(a32[i, 0]) == (-1, nid)
        offs = a16[i, 1 : cnt[i]].astype(np.int64)
        np.testing.assert_array_equal(np.stack([rel[lo + offs], dst[lo + offs]], 1), a32[i, 1 : cnt[i]], err_msg=f"row {i}")
        assert (a16[i, cnt[i] :] == np.uint16(-7 & 0xFFFF)).all()  # nothing written past the list
        n_trimmed += cnt[i] == m.act_dim
    assert n_trimmed > 0 or tag in ("default", "hist0", "hist2")


def test_u16_action_lists_on_hub_rows_of_the_bench_graph():
    """The ml-1m-shaped synthetic CKG of bench_pgpr.py (max degree ~6 300): rows that end on the most connected items and
    users go through the wave trim (<= 2 048 neighbours) and the block trim (<= 8 192), rows that end on quiet nodes through
    the first pass -- in both layouts, which must describe the same actions."""
    import bench_pgpr
    from hopwise_amd import pgpr_ops as P

    kg, ds, m = bench_pgpr.make_model(dev(), dropout="identity")
    g = m.graph
    deg = (g.ptr[1:] - g.ptr[:-1]).cpu().numpy()
    item_deg = deg[g.n_users : g.n_users + g.n_items]
    hubs = np.argsort(-item_deg)[:40]  # items, by degree
    quiet = np.argsort(item_deg)[-400:-360]
    busy_users = np.argsort(-deg[: g.n_users])[:20]
    assert item_deg[hubs[0]] > 2048 and (item_deg[hubs] <= 8192).all()
    rng = np.random.default_rng(0)
    users = rng.integers(1, g.n_users, 8)
    rows = []
    for b, u in enumerate(users):
        for it in np.concatenate([hubs, quiet]):  # user -> item
            rows.append([u, 2, g.ui_relation, 1, it, 0, 0, 0, 0, 0, 0, b])
        for v in busy_users:  # user -> hub item -> busy user
            if v != u:
                rows.append([u, 3, g.ui_relation, 1, hubs[0], g.ui_relation, 0, v, 0, 0, 0, b])
    rec = torch.tensor(rows, dtype=torch.int32, device=dev())
    n = rec.shape[0]
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev())
    su, se = P.user_scores(g, *m._t64, torch.from_numpy(users).to(dev()))
    scratch = torch.empty((min(2048, n), g.max_degree), dtype=torch.int64, device=dev())  # the launcher asks for it past 4 096
    a32 = torch.full((n, m.act_dim, 2), -7, dtype=torch.int32, device=dev())
    n32 = torch.zeros(n, dtype=torch.int32, device=dev())
    P.actions(g, su, se, m._score_rel_entity, rec, n_rows, n, m.max_acts, False, a32, n32, scratch)
    a16 = torch.full((n, P.act16_ld(m.act_dim)), -7, dtype=torch.int16, device=dev())
    n16 = torch.zeros(n, dtype=torch.int32, device=dev())
    P.actions(g, su, se, m._score_rel_entity, rec, n_rows, n, m.max_acts, False, a16, n16, scratch, u16=True)
    assert torch.equal(n16, n32)
    ptr, rel, dst = (x.cpu().numpy() for x in (g.ptr, g.rel, g.dst))
    a32, a16, cnt = a32.cpu().numpy(), a16.cpu().numpy().view(np.uint16), n32.cpu().numpy()
    classes = {"first pass": 0, "wave trim": 0, "block trim": 0}
    for i, r in enumerate(rows):
        t, nid = (r[3], r[4]) if r[1] == 2 else (r[6], r[7])
        gid = nid if t == 0 else g.n_users + nid
        lo, d = ptr[gid], deg[gid]
        offs = a16[i, 1 : cnt[i]].astype(np.int64)
        assert a16[i, 0] == 0xFFFF and (offs < d).all()
        np.testing.assert_array_equal(np.stack([rel[lo + offs], dst[lo + offs]], 1), a32[i, 1 : cnt[i]], err_msg=f"row {i}")
        classes["first pass" if cnt[i] < m.act_dim else ("wave trim" if d <= 2048 else "block trim")] += 1
    assert all(v > 0 for v in classes.values()), classes


def test_lds_mark_pass_equals_the_direct_one():
    """collect_mark_lds_kernel (a block's leaves maximised in LDS, touched cells flushed) against collect_mark_kernel (one
    memory-side atomic per leaf): same table, hence the same scores and selections bit for bit.  The switch is read once per
    process, so the direct form runs in a child process."""
    import os
    import subprocess
    import sys
    import tempfile

    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, ROOTDIR)\n"
        "from tests.test_gpu_pgpr import build_model, dev\n"
        "g, z, orc, m = build_model('default')\n"
        "users = torch.from_numpy(z['users']).to(dev())\n"
        "out = {}\n"
        "for compact in (False, True):\n"
        "    m._drop_calls = 0\n"
        "    res = m.beam_search_device(users, [64, 64, 64], fused_tail=True, compact_leaves=compact)\n"
        "    s, sel = m.collect_scores_device(users, res, n_nodes=4)\n"
        "    out['s%d' % compact] = s.cpu().numpy()\n"
        "    for k, v in sel.items(): out['%s%d' % (k, compact)] = v.cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n"
    ).replace("ROOTDIR", repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    got = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ("1", "0"):
            path = os.path.join(td, f"m{mode}.npz")
            env = dict(os.environ, HW_PGPR_MARK_LDS=mode)
            p = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300)
            assert p.returncode == 0, p.stderr[-2000:]
            got[mode] = dict(np.load(path))
    assert set(got["1"]) == set(got["0"]) and np.isfinite(got["1"]["s0"]).any()
    for k in got["1"]:
        np.testing.assert_array_equal(got["1"][k], got["0"][k], err_msg=k)


def test_duplicate_users_in_a_batch_compact_equals_full():
    """A user listed twice in a search: both copies expand (their leaves carry different batch indices) and the collector
    folds them into the first copy's slot -- the same through compact leaves and through full records."""
    g, z, orc, m = build_model("default")
    u = torch.from_numpy(z["users"]).to(dev())
    users = torch.cat([u[:5], u[2:4], u[:1]])
    for hops in ([25, 5, 1], [64, 64, 64]):
        full, (p0, q0) = _search(m, users, hops, fused_tail=True, compact_leaves=False)
        comp, (p1, q1) = _search(m, users, hops, fused_tail=True, compact_leaves=True)
        np.testing.assert_array_equal(p1, p0)
        np.testing.assert_array_equal(q1, q0)
        s0, sel0 = m.collect_scores_device(users, full, n_nodes=4)
        s1, sel1 = m.collect_scores_device(users, comp, n_nodes=4)
        np.testing.assert_array_equal(s1.cpu().numpy(), s0.cpu().numpy())
        for k in sel0:
            assert torch.equal(sel1[k], sel0[k]), (hops, k)
        cnt = sel0["count"].cpu().numpy()
        assert (cnt[5:] == 0).all() and cnt[:5].sum() > 0  # the later copies own no result row


def test_one_user_and_no_user():
    g, z, orc, m = build_model("default")
    u = torch.from_numpy(z["users"]).to(dev())
    one_f, (p0, q0) = _search(m, u[:1], [25, 5, 1], fused_tail=False)
    one_c, (p1, q1) = _search(m, u[:1], [25, 5, 1])
    assert one_c.leaves is not None and len(p0) > 0
    np.testing.assert_array_equal(p1, p0)
    np.testing.assert_allclose(q1, q0, rtol=3e-7, atol=0)  # k <= 8 hops: the two-launch form ranks with another kernel body
    none = m.beam_search_device(u[:0], [25, 5, 1])
    assert none.n == 0 and none.numpy()[0].shape == (0, 12)
    scores, sel = m.collect_scores_device(u[:1], one_c, n_nodes=4)
    assert tuple(scores.shape) == (1, int(g["n_items"]))


def test_collector_slot_and_row_numbering_on_a_large_batch_with_duplicates():
    """700 slots, users repeating across the 64- and 256-slot pieces the slot / row kernels work in: later copies own no result
    row, result rows number the slots that have one in order, and a first copy's row is the row of the same user searched in
    a batch without duplicates."""
    g, z, orc, m = build_model("default")
    rng = np.random.default_rng(11)
    pool = np.unique(z["users"])
    singles, repeated = pool[: len(pool) // 2], pool[len(pool) // 2 :]
    users_np = rng.choice(repeated, size=700, replace=True)
    users_np[rng.choice(700, size=len(singles), replace=False)] = singles  # these occur once, at scattered slots
    users = torch.from_numpy(users_np).to(dev())
    res = m.beam_search_device(users, [9, 3, 1])
    scores, sel = m.collect_scores_device(users, res, n_nodes=4)
    cnt, row = sel["count"].cpu().numpy(), sel["row"].cpu().numpy()
    first = {}
    for i, u in enumerate(users_np):
        first.setdefault(int(u), i)
    later = np.array([first[int(u)] != i for i, u in enumerate(users_np)])
    assert later.sum() > 300 and (cnt[later] == 0).all()
    want_row, r = [], 0
    for c in cnt:
        want_row.append(r if c > 0 else -1)
        r += c > 0
    np.testing.assert_array_equal(row, np.array(want_row))
    uniq = torch.from_numpy(np.array(sorted(first, key=first.get), np.int64)).to(dev())  # first occurrences, in slot order
    res_u = m.beam_search_device(uniq, [9, 3, 1])
    scores_u, sel_u = m.collect_scores_device(uniq, res_u, n_nodes=4)
    # dropout is off in this model: a user's beam does not depend on its position in the batch; only the FIRST copy's leaves
    # differ from the folded ones when copies exist, so compare users that occur once
    once = [i for i, u in enumerate(users_np) if (users_np == u).sum() == 1]
    assert len(once) == len(singles) > 3
    pos_u = {int(u): j for j, u in enumerate(uniq.cpu().numpy())}
    row_u = sel_u["row"].cpu().numpy()
    have = [i for i in once if row[i] >= 0]  # result rows are numbered over the slots that have one (sel["row"])
    assert len(have) > 2 and all(row_u[pos_u[int(users_np[i])]] >= 0 for i in have)
    a = scores.cpu().numpy()[row[have]]
    b = scores_u.cpu().numpy()[[row_u[pos_u[int(users_np[i])]] for i in have]]
    np.testing.assert_array_equal(a, b)


def test_collector_parent_walk_equals_the_record_form_on_a_large_search():
    """With 65 536 leaf slots or more (and an item table that fits LDS) the collector's best-path pass over compact leaves goes
    parent by parent (collect_mark_parents_kernel: the parent through the scalar unit, a lane per child); over full records it
    goes leaf by leaf.  Same scores, same selections, same explanation leaves."""
    g, z, orc, m = build_model("default")
    u = torch.from_numpy(z["users"]).to(dev())
    users = u.repeat(8)[:64]  # copies fold into the first one's slot on both paths
    assert m.act_dim >= 32
    hops = [32, 16, 8]
    full, (p0, q0) = _search(m, users, hops, fused_tail=True, compact_leaves=False)
    comp, (p1, q1) = _search(m, users, hops, fused_tail=True, compact_leaves=True)
    assert comp.leaves.shape[0] >= 65536 and comp.child_offsets is not None
    np.testing.assert_array_equal(p1, p0)
    s0, sel0 = m.collect_scores_device(users, full, n_nodes=4)
    s1, sel1 = m.collect_scores_device(users, comp, n_nodes=4)
    np.testing.assert_array_equal(s1.cpu().numpy(), s0.cpu().numpy())
    for k in sel0:
        assert torch.equal(sel1[k], sel0[k]), k
    assert int(sel0["count"].sum()) > 0
    # and without the child ranges the leaf-by-leaf form over the same compact leaves
    comp.child_offsets = None
    s2, sel2 = m.collect_scores_device(users, comp, n_nodes=4)
    np.testing.assert_array_equal(s2.cpu().numpy(), s0.cpu().numpy())
    for k in sel0:
        assert torch.equal(sel2[k], sel0[k]), k
