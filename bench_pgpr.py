#!/usr/bin/env python
"""PGPR paths expanded / s on BASELINE.json configs[2] (3-hop beam search, beam 64/64/64, ml-1m-small-shaped CKG).

Imported by bench.py (extra `pgpr` object in its JSON line) and runnable alone:
    python bench_pgpr.py [--users 32] [--steps 5] [--cpu-users 1]
paths expanded = sum over hops of the number of child paths emitted (pgpr.py:459-479); time = beam_search only.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

DIM = 100
HOPS = [64, 64, 64]


class _DS:
    """Synthetic KnowledgeBasedDataset stand-in with the attributes PGPR reads."""

    ui_relation = "[UI-Relation]"
    uid_field, iid_field = "user_id", "item_id"

    def __init__(self, kg, seed=2024):
        self.kg = kg
        self._num = dict(user_id=kg.n_users, item_id=kg.n_items, entity_id=kg.n_entities, relation_id=kg.n_relations)
        self.field2token_id = {"relation_id": {self.ui_relation: kg.ui_relation}}
        self.inter_feat = {"user_id": kg.inter_users, "item_id": kg.inter_items}
        self.head_entities, self.relations, self.tail_entities = kg.heads, kg.relations, kg.tails
        rng = np.random.default_rng(seed)
        s = 1.0 / np.sqrt(DIM)
        self._w = {
            "user_embedding_id": rng.standard_normal((kg.n_users, DIM)) * s,
            "entity_embedding_id": rng.standard_normal((kg.n_entities, DIM)) * s,
            "relation_embedding_id": rng.standard_normal((kg.n_relations, DIM)) * s,
        }

    def num(self, f):
        return self._num[f]

    def history_item_matrix(self):
        return np.zeros((self.kg.n_users, 1), np.int64), None, None

    def get_preload_weight(self, name):
        return self._w[name]


def make_model(dev, dropout="reference"):
    from hopwise_amd.data import synthetic
    from hopwise_amd.model.knowledge_aware_recommender import PGPR

    kg = synthetic.make("ml-1m-small", seed=2024, n_triples=178_000)
    ds = _DS(kg)
    cfg = dict(USER_ID_FIELD="user_id", ITEM_ID_FIELD="item_id", NEG_PREFIX="neg_", ENTITY_ID_FIELD="entity_id",
               RELATION_ID_FIELD="relation_id", HEAD_ENTITY_ID_FIELD="head_id", TAIL_ENTITY_ID_FIELD="tail_id",
               device=dev, topk=[10], state_history=1, max_acts=250, hidden_sizes=[512, 256], beam_search_hop=HOPS,
               fix_scores_sorting_bug=False, policy_dropout=dropout, seed=2024)
    torch.manual_seed(2024)
    m = PGPR(cfg, ds).to(dev)
    for lin in (m.l1, m.l2, m.actor, m.critic):
        torch.nn.init.xavier_normal_(lin.weight)
    return kg, ds, m


def algorithmic_bytes(m, res, hops):
    """Bytes THIS algorithm must move for one search, summed over the parent rows of every hop:
        CSR slice                       deg * 8          (int32 relation + int32 neighbour per edge)
        trim scores of over-full rows   deg * 16         (two float64 table entries per candidate instead of SURVEY's
                                                          deg * d * 4 gathered rows: the float64 trim tables u.U[n],
                                                          (u + r_ui).E[n], R[r].E[n] are precomputed per batch)
        first policy layer              n_parts * H1 * 4 (one 2 KB slot-table row per state slot; 4 at state_history 1)
        children                        min(k, n_acts) * 60   (48-byte path record + 3 float32 hop values); * 16 on the last
                                                          hop when it writes compact leaves {parent row, relation, node,
                                                          value} (BeamResult.leaves)
    `res` must come from beam_search_device(..., keep_buffers=True).  Returns (total, per-hop list)."""
    g, buf = m.graph, res.buffers
    nr = res.n_rows_per_hop.cpu().numpy()
    n_parts, h1 = 2 + 2 * m.state_history, m.hidden_sizes[0]
    total, per_hop = 0, []
    for hop, k in enumerate(hops):
        n = int(nr[hop])
        rec = buf["paths"][hop][:n].long()
        nn = rec[:, 1]  # nodes on the path so far
        col = 2 + 3 * (nn - 2).clamp_min(0)  # (rel, type, id) of the last node
        typ = torch.where(nn > 1, rec.gather(1, (col + 1).view(-1, 1)).view(-1), torch.zeros_like(nn))
        nid = torch.where(nn > 1, rec.gather(1, (col + 2).view(-1, 1)).view(-1), rec[:, 0])
        gid = torch.where(typ > 0, g.n_users + nid, nid)
        deg = g.ptr[gid + 1] - g.ptr[gid]
        # n_acts of this hop is overwritten by later hops in the shared buffer: children are bounded by k and by deg + 1
        kids = torch.minimum(torch.full_like(deg, k), torch.minimum(deg + 1, torch.full_like(deg, m.act_dim)))
        child_bytes = 16 if (hop == len(hops) - 1 and getattr(res, "leaves", None) is not None) else 60
        per_row = (deg * 8 + torch.where(deg > m.max_acts, deg * 16, torch.zeros_like(deg)) + n_parts * h1 * 4
                   + kids * child_bytes)
        b = int(per_row.sum().item())
        per_hop.append({"rows": n, "bytes": b})
        total += b
    return total, per_hop


def _cpu_workers(kg, ds, m, rng, n_workers, users_per_worker=2):
    """oracle/pgpr.py's beam search in ``n_workers`` child processes -> the cpu_baseline object, or None when a worker fails."""
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "pgpr_state.npz")
        W = {"w." + k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        np.savez(path, n_users=kg.n_users, n_entities=kg.n_entities, n_items=kg.n_items, ui_relation=kg.ui_relation,
                 inter_users=kg.inter_users, inter_items=kg.inter_items, heads=kg.heads, relations=kg.relations, tails=kg.tails,
                 user_emb=ds._w["user_embedding_id"], entity_emb=ds._w["entity_embedding_id"],
                 relation_emb=ds._w["relation_embedding_id"], max_acts=250, state_history=1, **W)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
        procs = []
        for _ in range(n_workers):
            users = [str(int(u)) for u in rng.integers(1, kg.n_users, users_per_worker)]
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.pgpr_worker", path, ",".join(map(str, HOPS)), *users],
                                          cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        res = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=300)
                res.append(json.loads(o.strip().splitlines()[-1]))
            except Exception:  # noqa: BLE001
                p.kill()
                return None
    paths, slowest = sum(r["paths"] for r in res), max(r["seconds"] for r in res)
    return {"value": paths / slowest, "unit": "paths/s", "cores": n_workers, "kind": "port",
            "sample": f"{n_workers} processes x {users_per_worker} users, one thread each, same graph/tables/MLP, oracle/pgpr.py "
                      f"beam_search; slowest worker {slowest:.1f} s"}


def run(dev, n_users=32, steps=0, warmup=1, cpu_users=0, rank=0, min_seconds=0.5, streams=1, cpu_workers=0):
    """steps = 0: as many searches as fill `min_seconds` of timed work (at least 5).  ``streams`` > 1: the timed searches are
    dealt round-robin over that many HIP streams (independent user batches: a search's policy launch is matrix-bound, its
    action / ranking launches are not, and kernels of two searches can share the chip)."""
    kg, ds, m = make_model(dev)
    rng = np.random.default_rng(100 + rank)
    expanded = 0

    def batch(i):
        return torch.from_numpy(rng.integers(1, kg.n_users, n_users)).to(dev)

    for i in range(max(warmup, 1)):
        res = m.beam_search_device(batch(i))
        torch.cuda.synchronize()
    if steps <= 0:  # sized from three back-to-back searches (a lone one also pays the launch-queue ramp), 10 % on top
        t0 = time.perf_counter()
        for i in range(3):
            m.beam_search_device(batch(i))
        torch.cuda.synchronize()
        steps = max(5, int(np.ceil(1.1 * min_seconds / max((time.perf_counter() - t0) / 3, 1e-4))))
    user_batches = [batch(i) for i in range(steps)]  # ids resident on the device before the clock starts
    counts = []
    # streams that were probed to sit on different hardware queues (hopwise_amd/stream_pool.py)
    from hopwise_amd.stream_pool import concurrent_streams

    side = concurrent_streams(streams, dev) if streams > 1 else None
    if side is not None:  # one untimed search per stream: its allocator pool is filled before the clock starts
        for st in side:
            with torch.cuda.stream(st):
                m.beam_search_device(user_batches[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if side is None:
            res = m.beam_search_device(user_batches[i])
        else:
            with torch.cuda.stream(side[i % streams]):
                res = m.beam_search_device(user_batches[i], overlap_actions=False)  # the searches fill each other's gaps
        counts.append(res.n_rows_per_hop)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    expanded = int(sum(int(c[1:].sum().item()) for c in counts))
    # byte accounting and per-stage times on more (untimed) searches of the same size
    res = m.beam_search_device(batch(steps), keep_buffers=True)
    abytes, per_hop = algorithmic_bytes(m, res, HOPS)
    # what the compact last level costs downstream, outside the timed region: the collector on it (reads the leaves in
    # place) and a rebuild of every leaf as a full 60-byte record (what nothing on the evaluation path asks for)
    leaf_layout = {"bytes_per_leaf": 16 if res.leaves is not None else 60}
    users_probe = batch(steps)
    res = m.beam_search_device(users_probe)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    m.collect_scores_device(users_probe, res)
    e[1].record()
    if res.leaves is not None:
        _ = res.paths  # materialises all leaves
    e[2].record()
    torch.cuda.synchronize()
    leaf_layout["collect_scores_ms"] = round(e[0].elapsed_time(e[1]), 4)
    if res.leaves is not None:
        leaf_layout["materialize_all_records_ms"] = round(e[1].elapsed_time(e[2]), 4)
    del res
    stage_ms = {}
    n_probe = 5
    for j in range(n_probe):
        evs = []
        m.beam_search_device(batch(steps + 1 + j), stage_events=evs)
        torch.cuda.synchronize()
        for (_, _, e0), (stage, hop, e1) in zip(evs[:-1], evs[1:]):
            key = f"{stage}.hop{hop + 1}"
            stage_ms[key] = stage_ms.get(key, 0.0) + e0.elapsed_time(e1) / n_probe
    by_stage = {}
    for k, v in stage_ms.items():
        by_stage[k.split(".")[0]] = by_stage.get(k.split(".")[0], 0.0) + v
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("pgpr_search_bytes_per_search")
        except Exception:
            traffic = None
    ms = dt / steps * 1e3
    # ---- roofline of the dominant kernel: the policy launch of the LAST hop (55-60 % of a search) -------------------------
    # It is matrix / issue bound, not bandwidth bound: per beam row two dense layers (H1 x 256 and 256 x 256 multiply-adds:
    # pgpr.py:175-187) evaluated as three-way bf16 splits -- six bf16 MFMAs per fp32 product, so the pipe's fp32 roof is the dense
    # bf16 rate / 6.  The slot-table rows it gathers are L2-served (PMC traffic of a whole search is BELOW the bytes the
    # algorithm addresses), so bytes are no lower bound for it; they are reported for the search as a whole (search_bytes).
    rows_last = int(counts[-1][len(HOPS) - 1].item()) if len(counts) else 0  # beam rows entering the last hop
    h1, h2 = m.hidden_sizes
    flops = 2.0 * rows_last * (h1 * h2 + h2 * 256)  # the actor layer is computed over 256 padded action slots
    pol_ms = stage_ms.get(f"policy.hop{len(HOPS)}", 0.0)
    mfma_peak = 2500.0 / 6.0
    tail_ms = stage_ms.get(f"expand.hop{len(HOPS)}", 0.0)
    n_leaves = int(counts[-1][len(HOPS)].item()) if len(counts) else 0
    roofline = {
        "bound": "mfma", "kernel": f"pgpr_policy_fused_kernel<{2 + 2 * m.state_history}, 3>, last hop ({rows_last} beam rows)",
        "achieved": (flops / (pol_ms * 1e-3) / 1e12) if pol_ms else 0.0, "peak": mfma_peak, "unit": "TFLOP/s",
        "frac": (flops / (pol_ms * 1e-3) / 1e12 / mfma_peak) if pol_ms else 0.0, "traffic": None,
        "kernel_ms": pol_ms, "flops_per_launch": flops,
        "peak_note": "dense bf16 MFMA 2 500 TFLOP/s / 6: every fp32 product is six bf16 MFMAs of the exact three-way splits "
                     "(fp32-level error, as the reference's fp32 policy network needs for its top-k cuts); kernel_ms from HIP "
                     "events around the launch on its stream (stage_ms_by_hop)",
        "leaf_writes": {"bound": "hbm", "kernel": "pgpr_topk_expand_kernel<true, true>, last hop", "kernel_ms": tail_ms,
                        "algorithmic_bytes_per_launch": 16.0 * n_leaves + 1280.0 * rows_last,
                        "achieved": ((16.0 * n_leaves + 1280.0 * rows_last) / (tail_ms * 1e-3) / 1e9) if tail_ms else 0.0,
                        "peak": 8000.0, "unit": "GB/s",
                        "frac": ((16.0 * n_leaves + 1280.0 * rows_last) / (tail_ms * 1e-3) / 1e9 / 8000.0) if tail_ms else 0.0,
                        "note": "16 B per leaf written + 1 KB of logits and <= 256 B of action offsets read per beam row; the "
                                "kernel is bound by vector-instruction issue (358 wave instructions per row: ranking), not by "
                                "these bytes (profiles/r4/pmc_pgpr_r4.txt)"},
        "search_bytes": {
            "achieved_GBps": abytes / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": abytes / (ms * 1e-3) / 1e9 / 8000.0,
            "algorithmic_bytes_per_search": abytes, "traffic": traffic, "per_hop": per_hop,
            "note": "a REPORTED figure, not a roofline: bytes this algorithm addresses (CSR slice deg * 8 + 2 float64 "
                    "trim-score lookups per candidate of an over-full row + one 2 KB slot-table row per state slot + 60 B per "
                    "child, 16 B per child of the last hop; bench_pgpr.algorithmic_bytes) over the wall time of the whole "
                    "search.  Counter traffic (fabric-side bytes of all kernels of one search, FETCH_SIZE / WRITE_SIZE passes) is "
                    "below it: the slot-table rows are served by L2"},
        "stage_ms": {k: round(v, 4) for k, v in sorted(by_stage.items())},
        "stage_ms_by_hop": {k: round(v, 4) for k, v in sorted(stage_ms.items())},
        "note": "stage_ms: HIP events between the stages of a search (5 untimed searches; with the one-launch tail the top-k has "
                "no launch of its own and is inside 'expand'); leaf_layout: the collector on the compact leaves and a full rebuild "
                "of all 60-byte records, timed once outside the timed region; per-kernel durations: profiles/<round>/"
                "pgpr_kernel_stats_*.csv",
    }
    out = {
        "metric": "PGPR paths expanded per second (3-hop beam 64/64/64, d=100, max_acts=250)",
        "value": expanded / dt, "unit": "paths/s", "users_per_step": n_users, "steps": steps, "timed_seconds": dt,
        "ms_per_step": ms, "paths_per_step": expanded / steps,
        "streams": streams,  # the timed searches (independent user batches) are dealt round-robin over this many HIP streams
        "graph": {"n_users": kg.n_users, "n_items": kg.n_items, "n_entities": kg.n_entities, "nnz": m.graph.nnz,
                  "max_degree": m.graph.max_degree},
        "policy": ("one launch per hop (hw_pgpr_policy_fused)" if m.fused_policy else "three launches per hop (lookup, l2, actor)"),
        "tail": ("softmax + top-k + expand in one launch per hop (hw_pgpr_topk_expand)" if m.fused_tail
                 else "top-k and expand as two launches per hop"),
        "leaf_layout": leaf_layout,
        "roofline": roofline,
    }
    if cpu_users > 0:
        from oracle import pgpr as op

        g = op.CKG(kg.n_users, kg.n_entities, kg.ui_relation, kg.inter_users, kg.inter_items, kg.heads, kg.relations, kg.tails)
        W = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        orc = op.PGPROracle(g, kg.n_items, ds._w["user_embedding_id"], ds._w["entity_embedding_id"],
                            ds._w["relation_embedding_id"], np.zeros((kg.n_users, 1), np.int64), W, 250, 1, False, (10,))
        users = rng.integers(1, kg.n_users, cpu_users)
        t0 = time.perf_counter()
        opaths, oprobs, safe, trace = op.safe_leaves(orc, users, HOPS, with_trace=True)
        cdt = time.perf_counter() - t0
        n = sum(sum(min(k, len(a)) for a in t["acts"]) for t, k in zip(trace, HOPS))
        out["cpu_baseline"] = {"value": n / cdt, "unit": "paths/s", "cores": 1, "kind": "port",
                               "sample": f"{cpu_users} user(s), same graph/tables/MLP, oracle/pgpr.py beam_search in {cdt:.1f} s"}
        if cpu_workers > 1:
            # the same oracle on several cores: child processes (numpy only, one BLAS thread each) take two users apiece; the
            # rate is all their paths over the slowest worker's time.  They start after the timed GPU region: the host is theirs.
            multi = _cpu_workers(kg, ds, m, rng, cpu_workers)
            if multi is not None:
                multi["single_core"] = out["cpu_baseline"]["value"]
                out["cpu_baseline"] = multi
        # full-size parity, outside the timed region: the same users through the device search with the policy dropout
        # switched to identity (the oracle has none), leaves compared with the rule of tests/test_gpu_pgpr.py
        _, _, m_id = make_model(dev, dropout="identity")
        m_id.load_state_dict(m.state_dict())
        gp, gq = m_id.beam_search_device(torch.from_numpy(users).to(dev)).numpy()
        got = {tuple(int(x) for x in r[:11]) for r in gp}  # without the record's last word, the user's index in the batch

        def rec_of(path):  # oracle path [(rel, type, id) x n] -> the first 11 ints of the device's record
            r = [int(path[0][2]), len(path)] + [0] * 9
            for j, (rel, typ, nid) in enumerate(path[1:]):
                r[2 + 3 * j : 5 + 3 * j] = [int(rel), int(typ), int(nid)]
            return tuple(r)

        okeys = [rec_of(p) for p in opaths]
        missing = [k for k in okeys if k not in got]
        unsafe = {k for k, s in zip(okeys, safe) if not s}
        stray = [k for k in missing if k not in unsafe]
        if stray:
            raise AssertionError(f"PGPR full-size parity: {len(stray)} oracle leaves missing from the device search "
                                 f"without a float32 tie at a cut, e.g. {stray[:2]}")
        out["parity"] = {"users": int(cpu_users), "oracle_leaves": len(set(okeys)), "device_leaves": len(got),
                         "common": len(set(okeys) & got), "missing_all_at_ties": True,
                         "safe_leaves": len(set(okeys) - unsafe),
                         "rule": "every oracle leaf the device did not emit has a hop whose selected value is within 2e-6 "
                                 "of the first value cut (oracle.pgpr.safe_leaves)"}
        del m_id
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=64)
    ap.add_argument("--steps", type=int, default=0, help="0: enough searches for 0.5 s of timed work")
    ap.add_argument("--cpu-users", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--cpu-workers", type=int, default=0, help="CPU baseline on this many processes (with --cpu-users)")
    a = ap.parse_args()
    print(json.dumps(run(torch.device("cuda", 0), a.users, a.steps, 1, a.cpu_users, streams=a.streams, cpu_workers=a.cpu_workers)))
