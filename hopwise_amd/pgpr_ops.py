"""torch-tensor wrappers over the PGPR entry points of the C ABI (include/hopwise_amd.h)."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import PgprGraph, check
from .ops import _dev, _ptr, _stream

ROW_INTS = 12
USER, ENTITY, SELF_LOOP = 0, 1, -1
ACT_LDS_KEYS = 4096


class DeviceCKG:
    """The collaborative KG as one device CSR (rows sorted by (relation, dst), multiplicity kept)."""

    def __init__(self, n_users, n_items, n_entities, ui_relation, inter_users, inter_items, heads, rels, tails, device,
                 ui_bidirectional=True):
        iu, ii = np.asarray(inter_users, np.int64), np.asarray(inter_items, np.int64)
        h, r, t = np.asarray(heads, np.int64), np.asarray(rels, np.int64), np.asarray(tails, np.int64)
        ui = np.full(len(iu), int(ui_relation), np.int64)
        # gid: users [0, n_users), entities [n_users, n_users + n_entities)
        if ui_bidirectional:
            src = np.concatenate([iu, n_users + ii, n_users + h, n_users + t])
            rel = np.concatenate([ui, ui, r, r])
            dst = np.concatenate([ii, iu, t, h])
        else:  # CAFE's graph: ckg_dict_graph(ui_bidirectional=False) has no item -> user edges (cafe.py:86)
            src = np.concatenate([iu, n_users + h, n_users + t])
            rel = np.concatenate([ui, r, r])
            dst = np.concatenate([ii, t, h])
        order = np.lexsort((dst, rel, src))
        src, rel, dst = src[order], rel[order], dst[order]
        n = int(n_users) + int(n_entities)
        ptr = np.zeros(n + 1, np.int64)
        np.add.at(ptr, src + 1, 1)
        np.cumsum(ptr, out=ptr)
        self.n_users, self.n_items, self.n_entities, self.ui_relation = int(n_users), int(n_items), int(n_entities), int(ui_relation)
        self.max_degree = int(np.diff(ptr).max()) if n > 0 else 0
        self.nnz = len(src)
        self.device = torch.device(device)
        self.h_ptr, self.h_rel, self.h_dst = ptr, rel.astype(np.int32), dst.astype(np.int32)  # host copies (path samplers)
        self.ptr = torch.from_numpy(ptr).to(self.device)
        self.rel = torch.from_numpy(rel.astype(np.int32)).to(self.device)
        self.dst = torch.from_numpy(dst.astype(np.int32)).to(self.device)
        g = PgprGraph()
        g.ptr, g.rel, g.dst = self.ptr.data_ptr(), self.rel.data_ptr(), self.dst.data_ptr()
        g.n_users, g.n_entities, g.n_items, g.ui_relation = self.n_users, self.n_entities, self.n_items, self.ui_relation
        g.max_degree = self.max_degree
        self.c = g


def host_neighbours(graph, ntype, nid, rel):
    """graph_dict[type][id][rel] of the reference as a slice of the host CSR (ids ascending, multiplicity kept)."""
    gid = int(nid) if ntype == "user" else graph.n_users + int(nid)
    lo, hi = graph.h_ptr[gid], graph.h_ptr[gid + 1]
    r = graph.h_rel[lo:hi]
    a, b = np.searchsorted(r, rel, "left"), np.searchsorted(r, rel, "right")
    return graph.h_dst[lo + a : lo + b]


def init_paths(users, paths, probs, n_rows):
    lib = _lib.load()
    users = _dev(users, torch.int64, "users").contiguous()
    check(lib.hw_pgpr_init_paths(_ptr(users), users.numel(), _ptr(paths), _ptr(probs), _ptr(n_rows), _stream()),
          "hw_pgpr_init_paths")


def rel_entity_scores(R64, E64):
    """score_rel_entity[r, n] = R[r] . E[n] in float64 (once per model)."""
    lib = _lib.load()
    out = torch.empty((R64.shape[0], E64.shape[0]), dtype=torch.float64, device=R64.device)
    check(lib.hw_pgpr_rel_entity_scores(_ptr(R64), _ptr(E64), R64.shape[0], E64.shape[0], R64.shape[1], _ptr(out),
                                        _stream()), "hw_pgpr_rel_entity_scores")
    return out


def user_scores(graph, U64, E64, R64, users, user_vec=None):
    """(score_user [B, n_users], score_entity [B, n_entities]) float64 for the root users of a search batch.
    ``user_vec`` [n_users, d] float64 replaces the [UI-Relation] row per user (TPRec)."""
    lib = _lib.load()
    users = _dev(users, torch.int64, "users").contiguous()
    B, d = users.numel(), U64.shape[1]
    ws = torch.empty(2 * B * d, dtype=torch.float64, device=users.device)
    su = torch.empty((B, graph.n_users), dtype=torch.float64, device=users.device)
    se = torch.empty((B, graph.n_entities), dtype=torch.float64, device=users.device)
    if user_vec is not None:
        if user_vec.dtype != torch.float64 or tuple(user_vec.shape) != (graph.n_users, d) or not user_vec.is_contiguous():
            raise ValueError("user_vec must be contiguous float64 [n_users, d]")
        check(lib.hw_pgpr_user_scores_tp(C.byref(graph.c), _ptr(U64), _ptr(E64), _ptr(R64), _ptr(user_vec), d, _ptr(users),
                                         B, _ptr(ws), _ptr(su), _ptr(se), _stream()), "hw_pgpr_user_scores_tp")
        return su, se
    check(lib.hw_pgpr_user_scores(C.byref(graph.c), _ptr(U64), _ptr(E64), _ptr(R64), d, _ptr(users), B, _ptr(ws),
                                  _ptr(su), _ptr(se), _stream()), "hw_pgpr_user_scores")
    return su, se


QUEUE_TRIM_ROWS = True  # False: hw_pgpr_actions without its work buffer (the trim kernels scan every row)


TRIM_BLOCK_MAX_DEG = 8192  # csrc/pgpr.hip: rows beyond it go to the radix-select kernel, which writes (relation, dst) pairs only


def act16_ld(act_dim):
    return (int(act_dim) + 7) & ~7  # HW_PGPR_ACT16_LD


def acts_u16_supported(graph, fix_bug):
    """Whether the action lists of a search can be uint16 edge offsets (hw_pgpr_actions flag bit 2)."""
    return not fix_bug and graph.max_degree <= TRIM_BLOCK_MAX_DEG


def actions(graph, score_user, score_entity, score_rel_entity, paths, n_rows, max_rows, max_acts, fix_bug, acts, n_acts,
            scratch=None, start_user_revisitable=False, u16=False):
    """``u16``: ``acts`` is an int16 [max_rows, act16_ld(max_acts + 1)] tensor that receives edge offsets into the current
    node's CSR slice (0xFFFF = self loop) instead of int32 (relation, dst) pairs -- read only by topk_expand(acts_u16=True)."""
    lib = _lib.load()
    if u16 and (acts.dtype != torch.int16 or acts.shape[-1] != act16_ld(max_acts + 1) or not acts.is_contiguous()):
        raise ValueError("u16 action lists: int16 [rows, act16_ld(act_dim)] contiguous")
    sb = ss = 0
    if scratch is not None:
        sb, ss = scratch.shape
    work = None
    if QUEUE_TRIM_ROWS:
        # queues of the rows that need the trim.  Allocated per call (torch's allocator hands the block back in stream
        # order): until round 4 ONE buffer per (device, row capacity) was cached here, which two searches running on two
        # streams then shared -- queues of one search overwritten by the other's, row numbers beyond its own row count, a
        # memory access fault with three streams (tests/test_gpu_pgpr_streams.py)
        work = torch.empty(int(lib.hw_pgpr_actions_workspace_ints(int(max_rows))), dtype=torch.int32, device=paths.device)
    check(
        lib.hw_pgpr_actions(C.byref(graph.c), _ptr(score_user), _ptr(score_entity), _ptr(score_rel_entity), _ptr(paths),
                            _ptr(n_rows), int(max_rows), int(max_acts),
                            int(bool(fix_bug)) | (2 if start_user_revisitable else 0) | (4 if u16 else 0),  # HW_PGPR_* flag bits
                            _ptr(acts), _ptr(n_acts), _ptr(scratch), sb, ss, _ptr(work), 0 if work is None else work.numel(),
                            _stream()),
        "hw_pgpr_actions",
    )


def state(graph, U32, E32, R32, state_history, paths, n_rows, max_rows, out):
    lib = _lib.load()
    check(
        lib.hw_pgpr_state(C.byref(graph.c), _ptr(U32), _ptr(E32), _ptr(R32), U32.shape[1], int(state_history), _ptr(paths),
                          _ptr(n_rows), int(max_rows), _ptr(out), _stream()),
        "hw_pgpr_state",
    )


def mlp_layer(a, w, bias, out, m_rows, m_max, elu, drop_mask=None, drop_mode=0, seed=0, offset=0):
    """out[:m, :n] = f(a[:m, :k] @ w.T + bias); a / out may be wider than k / n (leading dimension = stride(0))."""
    lib = _lib.load()
    n, k = w.shape
    tiles = ((int(m_max) + 127) // 128) * ((n + 127) // 128)
    if tiles < 256 and k >= 256:
        # few rows (the first hops of a beam search): give the GEMM split-K room, or a lone tile walks K at one load
        # latency per 16-deep step
        splits = min(max(768 // tiles, 2), k // 64)
        ws = torch.empty(splits * int(m_max) * n, dtype=torch.float32, device=a.device)  # per call: see actions()
        check(
            lib.hw_gemm_f32(_ptr(a), a.stride(0), 1, _ptr(w), k, 1, _ptr(out), out.stride(0), _ptr(m_rows), int(m_max), n, k,
                            _ptr(bias), int(bool(elu)), _ptr(drop_mask), int(drop_mode), C.c_uint64(seed),
                            C.c_uint64(offset), _ptr(ws), ws.numel(), _stream()),
            "hw_gemm_f32",
        )
        return
    check(
        lib.hw_mlp_layer(_ptr(a), a.stride(0), _ptr(w), _ptr(bias), _ptr(out), out.stride(0), _ptr(m_rows), int(m_max),
                         n, k, int(bool(elu)), _ptr(drop_mask), int(drop_mode), C.c_uint64(seed), C.c_uint64(offset),
                         _stream()),
        "hw_mlp_layer",
    )


def layer1_tables(tables32, w1, state_history):
    """The slot tables of hw_pgpr_layer1_lookup for first-layer weights ``w1`` [hidden, (2 + 2 h) * d]: (U, E, R) float32
    tables times the matching column block of w1, on the MFMA GEMM."""
    from .ops import gemm

    U, E, R = tables32
    d = U.shape[1]
    blk = lambda s: w1[:, s * d : (s + 1) * d]
    out = [gemm(U, blk(0)), gemm(U, blk(1)), gemm(E, blk(1))]
    if state_history >= 1:
        out += [gemm(U, blk(2)), gemm(E, blk(2)), gemm(R, blk(3))]
    if state_history >= 2:
        out += [gemm(U, blk(4)), gemm(E, blk(4)), gemm(R, blk(5))]
    return out


def layer1_lookup(slot_tables, bias, state_history, paths, n_rows, max_rows, out, drop_mask=None, drop_mode=0, seed=0,
                  offset=0):
    lib = _lib.load()
    arr = (C.c_void_p * len(slot_tables))(*[t.data_ptr() for t in slot_tables])
    check(
        lib.hw_pgpr_layer1_lookup(arr, _ptr(bias), bias.numel(), int(state_history), _ptr(paths), _ptr(n_rows),
                                  int(max_rows), _ptr(drop_mask), int(drop_mode), C.c_uint64(seed), C.c_uint64(offset),
                                  _ptr(out), out.stride(0), _stream()),
        "hw_pgpr_layer1_lookup",
    )


def policy_fused_supported(hidden_sizes, act_dim):
    """Shapes hw_pgpr_policy_fused covers (hopwise's PGPR.yaml default [512, 256], max_acts 250 is one)."""
    return (len(hidden_sizes) == 2 and hidden_sizes[1] == 256 and hidden_sizes[0] % 128 == 0 and
            128 <= hidden_sizes[0] <= 1024 and 1 <= act_dim <= 256)


def policy_prepare(w2, wa):
    """l2.weight [256, H1], actor.weight [act_dim, 256] -> the pre-split, pre-swizzled bf16 plane images the fused policy
    kernel streams through LDS (once per weight version)."""
    lib = _lib.load()
    w2 = w2.detach().contiguous()
    wa = wa.detach().contiguous()
    n = int(lib.hw_pgpr_policy_planes_bytes(w2.shape[1]))
    if n <= 0:
        raise _lib.HopwiseAmdError("hw_pgpr_policy_prepare: unsupported hidden size")
    planes = torch.empty(n, dtype=torch.uint8, device=w2.device)
    check(lib.hw_pgpr_policy_prepare(_ptr(w2), w2.shape[1], w2.shape[0], _ptr(wa), wa.shape[0], _ptr(planes), _stream()),
          "hw_pgpr_policy_prepare")
    return planes


def policy_fused(slot_tables, zero_row, b1, planes, b2, ba_padded, act_dim, state_history, paths, n_rows, max_rows, logits,
                 drop_mode=0, mask1=None, mask2=None, seed=0, offset1=0, offset2=0):
    """logits[row, :act_dim] of the whole policy network for every beam row, one launch (csrc/pgpr_policy.hip).
    ``logits`` is [max_rows, >= 256]."""
    lib = _lib.load()
    arr = (C.c_void_p * len(slot_tables))(*[t.data_ptr() for t in slot_tables])
    check(
        lib.hw_pgpr_policy_fused(arr, _ptr(zero_row), _ptr(b1), b1.numel(), _ptr(planes), _ptr(b2), _ptr(ba_padded),
                                 int(act_dim), int(state_history), _ptr(paths), _ptr(n_rows), int(max_rows), int(drop_mode),
                                 _ptr(mask1), _ptr(mask2), C.c_uint64(seed), C.c_uint64(offset1), C.c_uint64(offset2),
                                 _ptr(logits), logits.stride(0), _stream()),
        "hw_pgpr_policy_fused",
    )


def policy_topk(logits, act_dim, n_acts, n_rows, max_rows, k, out_idx, out_val, probs_out=None):
    lib = _lib.load()
    check(
        lib.hw_pgpr_policy_topk(_ptr(logits), logits.stride(0), int(act_dim), _ptr(n_acts), _ptr(n_rows), int(max_rows),
                                int(k), _ptr(out_idx), _ptr(out_val), _ptr(probs_out), _stream()),
        "hw_pgpr_policy_topk",
    )


def expand(graph, paths, probs, acts, act_dim, n_acts, n_rows, max_rows, topk_idx, topk_val, k, offsets, paths_out,
           probs_out, n_rows_out, capacity):
    lib = _lib.load()
    block_sums = torch.empty((int(max_rows) + 1023) // 1024 + 1, dtype=torch.int32, device=paths.device)
    check(
        lib.hw_pgpr_expand(C.byref(graph.c), _ptr(paths), _ptr(probs), _ptr(acts), int(act_dim), _ptr(n_acts), _ptr(n_rows),
                           int(max_rows), _ptr(topk_idx), _ptr(topk_val), int(k), _ptr(offsets), _ptr(block_sums),
                           _ptr(paths_out),
                           _ptr(probs_out), _ptr(n_rows_out), int(capacity), _stream()),
        "hw_pgpr_expand",
    )


def topk_expand_supported(act_dim, k):
    """Shapes hw_pgpr_topk_expand takes: one wave ranks a row's <= 256 logits in registers and owns its <= 64 children."""
    return int(act_dim) <= 256 and 1 <= int(k) <= 64


def topk_expand(graph, paths, probs, acts, act_dim, n_acts, n_rows, max_rows, logits, k, offsets, paths_out, probs_out,
                leaf_out, n_rows_out, capacity, acts_u16=False):
    """Masked softmax + top-k + expand of one hop in a single launch (csrc/pgpr.hip: pgpr_topk_expand_kernel); the same
    children, in the same order, as policy_topk followed by expand.  ``leaf_out`` (int32 [capacity, 4]) selects the compact
    output of the last hop: {parent row, relation, node, value bits} instead of a 12-int record + probability triple."""
    lib = _lib.load()
    block_sums = torch.empty((int(max_rows) + 1023) // 1024 + 1, dtype=torch.int32, device=paths.device)
    check(
        lib.hw_pgpr_topk_expand(C.byref(graph.c), _ptr(paths), _ptr(probs), _ptr(acts), int(act_dim), _ptr(n_acts),
                                _ptr(n_rows), int(max_rows), _ptr(logits), logits.stride(0), int(k), _ptr(offsets),
                                _ptr(block_sums), _ptr(paths_out), _ptr(probs_out), _ptr(leaf_out), _ptr(n_rows_out),
                                int(capacity), 1 if acts_u16 else 0, _stream()),
        "hw_pgpr_topk_expand",
    )


def materialize_leaves(graph, parent_paths, parent_probs, leaves, n_leaf, sel=None, n_out=None):
    """Compact leaves -> (paths int32 [n, 12], probs float32 [n, 3]).  ``sel`` (int32 tensor): only those leaves, in that
    order; otherwise ``n_out`` rows (default: all of the buffer; rows past n_leaf[0] are left unwritten)."""
    lib = _lib.load()
    dev = leaves.device
    if sel is not None:
        sel = sel.to(device=dev, dtype=torch.int32).contiguous()
        n = sel.numel()
    else:
        n = leaves.shape[0] if n_out is None else int(n_out)
    paths = torch.empty((n, ROW_INTS), dtype=torch.int32, device=dev)
    probs = torch.empty((n, 3), dtype=torch.float32, device=dev)
    check(
        lib.hw_pgpr_materialize_leaves(C.byref(graph.c), _ptr(parent_paths), _ptr(parent_probs), _ptr(leaves), _ptr(n_leaf),
                                       n if sel is None else leaves.shape[0], _ptr(sel), n if sel is not None else 0,
                                       _ptr(paths), _ptr(probs), _stream()),
        "hw_pgpr_materialize_leaves",
    )
    return paths, probs


def positives_bitset(positives, n_users, n_items, device):
    """Bitset [n_users, ceil(n_items / 32)] of the rows of dataset.history_item_matrix()[0] (zeros included as they
    appear: `pid in self.positives[uid]` in the reference is a plain membership test on the padded row)."""
    pos = positives.cpu().numpy() if isinstance(positives, torch.Tensor) else np.asarray(positives)
    words = (int(n_items) + 31) // 32
    bits = np.zeros((int(n_users), words), np.uint32)
    rows = np.repeat(np.arange(pos.shape[0]), pos.shape[1])
    vals = pos.reshape(-1).astype(np.int64)
    ok = (vals >= 0) & (vals < n_items) & (rows < n_users)
    np.bitwise_or.at(bits, (rows[ok], vals[ok] >> 5), (np.uint32(1) << (vals[ok] & 31).astype(np.uint32)))
    return torch.from_numpy(bits.view(np.int32)).to(device)


def collect(graph, U64, E64, R64, users, pos_bits, leaf_paths, leaf_probs, n_leaf, max_leaf, n_nodes, kmax, leaves=None,
            child_offsets=None, n_parents=None):
    """``leaves`` (int32 [max_leaf, 4]): the compact output of the last hop; leaf_paths / leaf_probs are then the PARENT rows
    (hw_pgpr_collect_leaves).  ``child_offsets`` (int32, first leaf of every parent row: topk_expand's offsets of the last
    hop) + ``n_parents`` (device int32 [1]): the best-path pass goes parent by parent (hw_pgpr_collect_leaves_ex)."""
    lib = _lib.load()
    dev = users.device
    B, nI = users.numel(), graph.n_items
    best = torch.empty(B * nI, dtype=torch.int64, device=dev)
    score_ws = torch.empty(B * nI, dtype=torch.float64, device=dev)
    first = torch.empty(B, dtype=torch.int32, device=dev)
    results = torch.empty((B, nI), dtype=torch.float32, device=dev)
    sel_item = torch.empty((B, kmax), dtype=torch.int32, device=dev)
    sel_score = torch.empty((B, kmax), dtype=torch.float64, device=dev)
    sel_leaf = torch.empty((B, kmax), dtype=torch.int32, device=dev)
    sel_count = torch.empty(B, dtype=torch.int32, device=dev)
    sel_row = torch.empty(B, dtype=torch.int32, device=dev)
    if leaves is not None and child_offsets is not None and n_parents is not None:
        check(
            lib.hw_pgpr_collect_leaves_ex(C.byref(graph.c), _ptr(U64), _ptr(E64), _ptr(R64), U64.shape[1], _ptr(users), B,
                                          _ptr(pos_bits), pos_bits.shape[1], _ptr(leaf_paths), _ptr(leaf_probs),
                                          _ptr(leaves), _ptr(n_leaf), int(max_leaf), int(n_nodes), int(kmax), _ptr(best),
                                          _ptr(score_ws), _ptr(first), _ptr(results), _ptr(sel_item), _ptr(sel_score),
                                          _ptr(sel_leaf), _ptr(sel_count), _ptr(sel_row), _ptr(child_offsets),
                                          _ptr(n_parents), int(leaf_paths.shape[0]), _stream()),
            "hw_pgpr_collect_leaves_ex",
        )
        return results, dict(item=sel_item, score=sel_score, leaf=sel_leaf, count=sel_count, row=sel_row)
    if leaves is not None:
        check(
            lib.hw_pgpr_collect_leaves(C.byref(graph.c), _ptr(U64), _ptr(E64), _ptr(R64), U64.shape[1], _ptr(users), B,
                                       _ptr(pos_bits), pos_bits.shape[1], _ptr(leaf_paths), _ptr(leaf_probs), _ptr(leaves),
                                       _ptr(n_leaf), int(max_leaf), int(n_nodes), int(kmax), _ptr(best), _ptr(score_ws),
                                       _ptr(first), _ptr(results), _ptr(sel_item), _ptr(sel_score), _ptr(sel_leaf),
                                       _ptr(sel_count), _ptr(sel_row), _stream()),
            "hw_pgpr_collect_leaves",
        )
        return results, dict(item=sel_item, score=sel_score, leaf=sel_leaf, count=sel_count, row=sel_row)
    check(
        lib.hw_pgpr_collect(C.byref(graph.c), _ptr(U64), _ptr(E64), _ptr(R64), U64.shape[1], _ptr(users), B, _ptr(pos_bits),
                            pos_bits.shape[1], _ptr(leaf_paths), _ptr(leaf_probs), _ptr(n_leaf), int(max_leaf),
                            int(n_nodes), int(kmax), _ptr(best), _ptr(score_ws), _ptr(first), _ptr(results),
                            _ptr(sel_item), _ptr(sel_score), _ptr(sel_leaf), _ptr(sel_count), _ptr(sel_row), _stream()),
        "hw_pgpr_collect",
    )
    return results, dict(item=sel_item, score=sel_score, leaf=sel_leaf, count=sel_count, row=sel_row)


def log_softmax_rows(x, n_cols=None, out=None):
    lib = _lib.load()
    x = _dev(x, torch.float32, "x")
    n_cols = x.shape[1] if n_cols is None else n_cols
    if out is None:
        out = torch.empty_like(x)
    check(lib.hw_log_softmax_rows(_ptr(x), x.stride(0), x.shape[0], int(n_cols), _ptr(out), out.stride(0), _stream()),
          "hw_log_softmax_rows")
    return out


def _zeros_carved(dev, *specs):
    """Several zero-filled int32 / float32 tensors carved out of ONE zeroed buffer (one fill launch instead of one per
    tensor: a run_program call made ten, 45 us of its 750 us of device work).  specs: (shape, dtype); pieces start on
    16-byte boundaries; a zero int32 word is a zero float32."""
    sizes = [int(np.prod(shape)) for shape, _ in specs]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 3) & ~3
    buf = torch.zeros(max(total, 1), dtype=torch.int32, device=dev)
    out = []
    for (shape, dtype), n, o in zip(specs, sizes, offs):
        t = buf[o : o + n]
        if dtype == torch.float32:
            t = t.view(torch.float32)
        out.append(t.view(shape))
    return out


def cafe_execute(graph, nodes_np, node_ptr_np, users, logprobs, excluded_bits, grow_excluded, slot_ptr_np, n_slots,
                 keep_tables=False, on_device=None, row_lse=None):
    """Run the program interpreter.  nodes_np: structured array of CafeNode fields; returns (ids [n_slots, 3] int32,
    vals [n_slots, 3] float32, counts [n_nodes] int32) on the device.  ``row_lse`` [rows, 2] (``ops.gemm_lse``): ``logprobs``
    holds the logits and an entry's log-probability is taken where it is read (``hw_cafe_execute_logits``)."""
    lib = _lib.load()
    dev = users.device
    if on_device is not None:  # (nodes [n, 10] int32, node_ptr int32, slot_ptr int64) already copied by the caller
        nodes, node_ptr, slot_ptr = on_device
    else:
        nodes = torch.from_numpy(nodes_np.view(np.int32).reshape(-1, 10).copy()).to(dev)
        node_ptr = torch.from_numpy(np.asarray(node_ptr_np, np.int32)).to(dev)
        slot_ptr = torch.from_numpy(np.asarray(slot_ptr_np, np.int64)).to(dev)
    ids, vals, counts = _zeros_carved(dev, ((max(n_slots, 1), 3), torch.int32), ((max(n_slots, 1), 3), torch.float32),
                                      ((max(len(nodes_np), 1),), torch.int32))
    max_nodes = int(np.diff(node_ptr_np).max()) if len(node_ptr_np) > 1 else 0
    if row_lse is not None:
        if row_lse.dim() != 2 or row_lse.shape[1] != 2 or not row_lse.is_contiguous() or row_lse.shape[0] < logprobs.shape[0]:
            raise ValueError("row_lse: contiguous [rows, 2]")
        check(
            lib.hw_cafe_execute_logits(C.byref(graph.c), _ptr(nodes), _ptr(node_ptr), _ptr(users), users.numel(), max_nodes,
                                       _ptr(logprobs), logprobs.stride(0), _ptr(row_lse), _ptr(excluded_bits),
                                       excluded_bits.shape[1], int(bool(grow_excluded)), _ptr(ids), _ptr(vals), _ptr(counts),
                                       _ptr(slot_ptr), _stream()),
            "hw_cafe_execute_logits",
        )
    else:
        check(
            lib.hw_cafe_execute(C.byref(graph.c), _ptr(nodes), _ptr(node_ptr), _ptr(users), users.numel(), max_nodes,
                                _ptr(logprobs), logprobs.stride(0), _ptr(excluded_bits), excluded_bits.shape[1],
                                int(bool(grow_excluded)), _ptr(ids), _ptr(vals), _ptr(counts), _ptr(slot_ptr), _stream()),
            "hw_cafe_execute",
        )
    if keep_tables:
        return ids, vals, counts, nodes, node_ptr, slot_ptr
    return ids, vals, counts


def cafe_plan(order_of_task, tables, n_tasks, k):
    """hw_cafe_plan: the node table, pointers, leaf rows and module-stage index columns of a run_program batch from each task's
    visiting order.  ``tables``: the device dictionary CAFE._order_tables_dev builds.  Returns a dict of device tensors."""
    lib = _lib.load()
    dev = order_of_task.device
    L, n_leaves = tables["L"], tables["n_leaves"]
    total = n_tasks * L
    nodes = torch.empty((max(total, 1), 10), dtype=torch.int32, device=dev)
    ptrs = torch.empty(n_tasks + 2 + ((n_tasks + 1) & 1), dtype=torch.int32, device=dev)  # node_ptr [B + 1], padded to 8 bytes
    i64 = torch.empty(n_tasks + 1 + 3 * max(total, 1), dtype=torch.int64, device=dev)
    slot_ptr, idx = i64[: n_tasks + 1], i64[n_tasks + 1:]
    leaf = torch.empty((max(n_tasks, 1), n_leaves), dtype=torch.int32, device=dev)
    check(lib.hw_cafe_plan(_ptr(order_of_task), n_tasks, L, n_leaves, _ptr(tables["par"]), _ptr(tables["rel"]),
                           _ptr(tables["depth"]), _ptr(tables["cap"]), _ptr(tables["is_leaf"]), _ptr(tables["leaf"]),
                           tables["n_orders"], _ptr(tables["mod_rid"]), _ptr(tables["mod_eh"]), _ptr(tables["mod_vocab"]),
                           tables["n_mods"], _ptr(tables["key_first"]), _ptr(tables["key_count"]), tables["n_keys"], int(k),
                           tables["slots_per_task"], _ptr(nodes), _ptr(ptrs), _ptr(slot_ptr), _ptr(idx[:total]),
                           _ptr(idx[total: 2 * total]), _ptr(idx[2 * total: 3 * total]), _ptr(leaf), _stream()), "hw_cafe_plan")
    return dict(nodes=nodes, node_ptr=ptrs[: n_tasks + 1], slot_ptr=slot_ptr, idx=idx, leaf_nodes=leaf)


def cafe_modules_forward(mod_structs, group_mod, group_ptr, user_vec, eh_idx, uv_idx, out_rows, vec_all):
    """hw_cafe_modules_forward: every (level, relation) module application of a run_program call from one host call.
    mod_structs: ctypes array of _lib.CafeModule; group_mod / group_ptr: host sequences; index columns and vectors on the device."""
    lib = _lib.load()
    n_groups = len(group_mod)
    gm = (C.c_int32 * max(n_groups, 1))(*[int(x) for x in group_mod])
    gp = (C.c_int64 * (n_groups + 1))(*[int(x) for x in group_ptr])
    d = user_vec.shape[1]
    max_rows = max([gp[g + 1] - gp[g] for g in range(n_groups)] + [0])
    n_ws = int(lib.hw_cafe_modules_workspace_floats(max_rows, d))
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=user_vec.device)
    check(lib.hw_cafe_modules_forward(mod_structs, gm, gp, n_groups, _ptr(user_vec), user_vec.shape[0], d, _ptr(eh_idx),
                                      _ptr(uv_idx), _ptr(out_rows), _ptr(vec_all), _ptr(ws), n_ws, _stream()),
          "hw_cafe_modules_forward")


def cafe_modules_forward_fused(mod_structs, group_mod, group_ptr, group_level, user_vec, eh_idx, uv_idx, out_rows, vec_all):
    """hw_cafe_modules_forward_fused: one launch per level; ``mod_structs`` carries TRANSPOSED weights.  Returns False when the
    library refuses the shape (2 * dim > 256): the caller then takes hw_cafe_modules_forward."""
    lib = _lib.load()
    n_groups = len(group_mod)
    gm = (C.c_int32 * max(n_groups, 1))(*[int(x) for x in group_mod])
    gp = (C.c_int64 * (n_groups + 1))(*[int(x) for x in group_ptr])
    gl = (C.c_int32 * max(n_groups, 1))(*[int(x) for x in group_level])
    rc = lib.hw_cafe_modules_forward_fused(mod_structs, gm, gp, gl, n_groups, _ptr(user_vec), user_vec.shape[0], user_vec.shape[1],
                                           _ptr(eh_idx), _ptr(uv_idx), _ptr(out_rows), _ptr(vec_all), _stream())
    if rc == -2:  # HW_ERR_UNSUPPORTED
        return False
    check(rc, "hw_cafe_modules_forward_fused")
    return True


def cafe_overflow_check(rec):
    """Raise if hw_cafe_collect met a user whose paths reach more distinct items than it ranks (reads one int from the device:
    a sync).  Out of reach with the reference's manual_topk = 5 (at most 125 paths per leaf, 64 leaves: 8 000 entries)."""
    seen = int(rec["overflow"].item())
    if seen:
        raise RuntimeError(f"hw_cafe_collect: a user's paths reach {seen} distinct items, more than the 8192 one workgroup "
                           "ranks in LDS (manual_topk 5 gives at most 125 paths per metapath).  Lower manual_topk, or run the "
                           "users of such a layout in several run_program calls over subsets of the metapaths")


def cafe_collect(raw, leaf_nodes_np, n_items, max_rec, check_overflow=True, leaf_nodes_dev=None):
    """collect_results + ranking of run_program on the interpreter's device buffers (hw_cafe_collect).  ``raw``: what
    CAFE._run_programs(..., raw=True) returns; leaf_nodes_np int32 [n_tasks, n_leaves].  Returns (results [n_tasks, n_items]
    float32 on the device, dict of record tensors: item / score / leaf [n_tasks, max_rec], ids [n_tasks, max_rec, 3], count)."""
    lib = _lib.load()
    dev = raw["ids"].device
    n_tasks, n_leaves = leaf_nodes_np.shape
    leaf_nodes = (leaf_nodes_dev if leaf_nodes_dev is not None
                  else torch.from_numpy(np.ascontiguousarray(leaf_nodes_np, np.int32)).to(dev))
    best = torch.empty((n_tasks, n_items), dtype=torch.int64, device=dev)
    last = torch.empty((n_tasks, n_items), dtype=torch.int32, device=dev)
    results = torch.empty((n_tasks, n_items), dtype=torch.float32, device=dev)
    item, score, leaf, ids3, count, overflow = _zeros_carved(
        dev, ((n_tasks, max_rec), torch.int32), ((n_tasks, max_rec), torch.float32), ((n_tasks, max_rec), torch.int32),
        ((n_tasks, max_rec, 3), torch.int32), ((n_tasks,), torch.int32), ((1,), torch.int32))
    rec = dict(item=item, score=score, leaf=leaf, ids=ids3, count=count)
    check(
        lib.hw_cafe_collect(_ptr(raw["nodes"]), _ptr(raw["node_ptr"]), _ptr(raw["slot_ptr"]), _ptr(raw["ids"]), _ptr(raw["vals"]),
                            _ptr(raw["counts"]), _ptr(leaf_nodes), n_tasks, n_leaves, int(n_items), _ptr(best), _ptr(last),
                            _ptr(results), int(max_rec), _ptr(rec["item"]), _ptr(rec["score"]), _ptr(rec["leaf"]),
                            _ptr(rec["ids"]), _ptr(rec["count"]), _ptr(overflow), _stream()),
        "hw_cafe_collect",
    )
    rec["overflow"] = overflow
    # check_overflow=False (run_program_device): no device sync here -- the scores of an overflowing user are NaN, the counter
    # travels with the records and cafe_overflow_check reads it when somebody copies results to the host anyway
    if check_overflow:
        cafe_overflow_check(rec)
    return results, rec


CAFE_NODE_DTYPE = np.dtype([(n, np.int32) for n in
                            ("parent", "rel", "eh_is_entity", "k", "is_leaf", "lp_row", "depth", "vocab", "cap", "slot_off")])


# ---- policy-gradient training (csrc/pgpr_train.hip) ---------------------------------------------------------------
def action_dropout(n_acts, n_rows, max_rows, act_dim, dropout, seed, offset, mask):
    lib = _lib.load()
    check(
        lib.hw_pgpr_action_dropout(_ptr(n_acts), _ptr(n_rows), int(max_rows), int(act_dim), float(dropout),
                                   C.c_uint64(seed), C.c_uint64(offset), _ptr(mask), _stream()),
        "hw_pgpr_action_dropout",
    )


def policy_sample(head, act_dim, mask, n_rows, max_rows, probs, acts, logp, entropy, value, seed=0, offset=0,
                  forced_acts=None):
    lib = _lib.load()
    check(
        lib.hw_pgpr_policy_sample(_ptr(head), head.stride(0), int(act_dim), _ptr(mask), _ptr(n_rows), int(max_rows),
                                  _ptr(forced_acts), C.c_uint64(seed), C.c_uint64(offset), _ptr(probs), _ptr(acts),
                                  _ptr(logp), _ptr(entropy), _ptr(value), _stream()),
        "hw_pgpr_policy_sample",
    )


def policy_grad(probs, act_dim, acts, coef_actor, coef_entropy, dvalue, n_rows, dhead, upstream=None):
    """``upstream``: float32 [1] on the device, the gradient arriving at the scalar loss (None = 1): dhead comes out scaled."""
    lib = _lib.load()
    check(
        lib.hw_pgpr_policy_grad_scaled(_ptr(probs), int(act_dim), _ptr(acts), _ptr(coef_actor), float(coef_entropy),
                                       _ptr(dvalue), int(n_rows), _ptr(dhead), dhead.stride(0), _ptr(upstream), _stream()),
        "hw_pgpr_policy_grad_scaled",
    )


def act_dropout_backward(dy, y, act, drop_mask=None, drop_mode=0, seed=0, offset=0, out=None):
    """dx for y = dropout(act(pre)); dy / y contiguous [m, n]; act: 0 none, 1 ELU, 2 ReLU."""
    lib = _lib.load()
    if not (dy.is_contiguous() and y.is_contiguous()):
        raise ValueError("act_dropout_backward wants contiguous operands")
    out = torch.empty_like(dy) if out is None else out
    check(
        lib.hw_act_dropout_backward(_ptr(dy), _ptr(y), dy.shape[0], dy.shape[1], int(act), _ptr(drop_mask),
                                    int(drop_mode), C.c_uint64(seed), C.c_uint64(offset), _ptr(out), _stream()),
        "hw_act_dropout_backward",
    )
    return out


def colsum(x, out=None):
    lib = _lib.load()
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device) if out is None else out
    n_ws = int(lib.hw_colsum_workspace_floats(x.shape[0], x.shape[1]))
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=x.device)
    check(lib.hw_colsum_f32_ws(_ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(out), _ptr(ws), n_ws, _stream()),
          "hw_colsum_f32_ws")
    return out


def reward(paths, n_rows, patterns, score_entity, n_items, scales, compute_scales, out):
    lib = _lib.load()
    check(
        lib.hw_pgpr_reward(_ptr(paths), int(n_rows), _ptr(patterns), 0 if patterns is None else patterns.shape[0],
                           _ptr(score_entity), score_entity.shape[1], int(n_items), _ptr(scales), score_entity.shape[0],
                           int(bool(compute_scales)), _ptr(out), _stream()),
        "hw_pgpr_reward",
    )


def reinforce_loss(rewards, values, logp, entropy, n_steps, n_batch, gamma, weight_factor, loss_out, coef_actor, dvalue):
    lib = _lib.load()
    check(
        lib.hw_pgpr_reinforce_loss(_ptr(rewards), _ptr(values), _ptr(logp), _ptr(entropy), int(n_steps), int(n_batch),
                                   float(gamma), float(weight_factor), _ptr(loss_out), _ptr(coef_actor), _ptr(dvalue),
                                   _stream()),
        "hw_pgpr_reinforce_loss",
    )
