"""torch-tensor wrappers over the C ABI (device pointers + current HIP stream) and their autograd glue.

PyTorch only provides device memory, the stream and autograd bookkeeping here; all arithmetic on
the path runs in the hand-written gfx950 kernels of ``csrc/``.  CPU tensors are rejected -- there
is deliberately no fallback implementation.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import MODEL_IDS, KgeDesc, KgeGrads, check

PAIRWISE_EPS = 1e-6  # torch pairwise_distance eps used by TripletMarginLoss (TransE loss path)

_N_SLOT = {"TransE": (1, 1, 1), "DistMult": (1, 1, 1), "RotatE": (2, 1, 2), "ComplEx": (2, 2, 2), "RESCAL": (1, 1, 1),
           "TorusE": (1, 1, 1), "TransH": (1, 2, 1), "TransD": (2, 2, 2), "Analogy": (3, 3, 3), "HolE": (1, 1, 1)}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object
    through several Python layers (~10 us; five of them per training step); the raw query is one C call."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor, dtype, what: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HopwiseAmdError(f"{what}: expected a CUDA(HIP) tensor; hopwise_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_DESC_CACHE = {}


def make_desc(model: str, head, rel, tail, margin: float = 0.0, eps: float = 0.0, tail_rows=None) -> KgeDesc:
    """Describe one scorer call.  ``head``/``rel``/``tail`` are tuples of [rows, dim] fp32 tables.
    ``tail_rows`` (optional) narrows the tail id space to [0, tail_rows) -- e.g. n_items on the rec side, where
    tails are items, the first rows of the entity table; ids are clamped to it and the launcher sizes its
    cache strategy from it."""
    nh, nr, nt = _N_SLOT[model]
    if (len(head), len(rel), len(tail)) != (nh, nr, nt):
        raise ValueError(f"{model} takes {nh}/{nr}/{nt} head/rel/tail tables")
    # the descriptor of the same tables is reused (nobody writes into one after this function): an evaluation step builds three
    # of them, ~13 us of ctypes field stores each, on a path that is bound by its launching thread at ml-1m's item count
    key = (model, tuple((t.data_ptr(), t.shape, t.dtype, t.is_contiguous()) for t in (*head, *rel, *tail)), float(margin),
           float(eps), tail_rows)
    hit = _DESC_CACHE.get(key)
    if hit is not None:
        return hit
    d = KgeDesc()
    d.model = MODEL_IDS[model]
    dim = head[0].shape[1]
    for slot, tabs in (("head", head), ("rel", rel), ("tail", tail)):
        arr = getattr(d, slot)
        for k in range(_lib.MAX_TABLES):
            arr[k] = None
        rows = tabs[0].shape[0]
        for k, t in enumerate(tabs):
            _dev(t, torch.float32, f"{slot} table")
            want = dim * dim if (model == "RESCAL" and slot == "rel") else dim  # RESCAL relation rows are d x d matrices
            if t.dim() != 2 or t.shape[1] != want or t.shape[0] != rows or not t.is_contiguous():
                raise ValueError(f"{slot} table {k}: need contiguous [rows, {want}]")
            arr[k] = t.data_ptr()
    d.dim = dim
    d.n_head, d.n_rel, d.n_tail = head[0].shape[0], rel[0].shape[0], tail[0].shape[0]
    if tail_rows is not None:
        if not 1 <= int(tail_rows) <= tail[0].shape[0]:
            raise ValueError("tail_rows out of range")
        d.n_tail = int(tail_rows)
    d.margin = float(margin)
    d.eps = float(eps)
    if len(_DESC_CACHE) >= 256:
        _DESC_CACHE.clear()
    _DESC_CACHE[key] = d
    return d


def _idx(t, what):
    t = _dev(t, torch.int64, what)
    return t


class _OobCounter:
    """Per-device int32 counter the kernels bump when an id falls outside its table."""

    _cache = {}

    @classmethod
    def get(cls, device):
        key = (device.type, device.index)
        if key not in cls._cache:
            cls._cache[key] = torch.zeros(1, dtype=torch.int32, device=device)
            # every stream of the process bumps this one word from now on: its zero fill must have landed before a kernel
            # on ANOTHER stream can touch it (once per process and device; not possible, nor needed, inside a capture --
            # the warm-up steps in front of a capture have created the counter already)
            if device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(device).synchronize()
        return cls._cache[key]


def oob_count(device=None) -> int:
    """Number of out-of-range ids seen so far on ``device`` (synchronises)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = int(_OobCounter.get(device).item())
    from . import torch_ops

    if torch_ops._state["loaded"]:  # the registered ops keep their own counter
        n += int(torch.ops.hopwise_amd.oob_count(_OobCounter.get(device)))
    return n


def kge_score_grouped_raw(desc, h_idx, r_idx, r_fixed, t_idx, j_major_out=False):
    """scores[b, j] for t_idx[b, j] (any strides).  Returns a [B, G] tensor; with ``j_major_out`` its
    memory is [G, B] contiguous (positives-first layout used by the BCE label convention)."""
    lib = _lib.load()
    h_idx = _idx(h_idx, "h_idx").contiguous()
    t_idx = _idx(t_idx, "t_idx")
    if t_idx.dim() != 2 or t_idx.shape[0] != h_idx.shape[0]:
        raise ValueError("t_idx must be [n_groups, n_tail_per_group]")
    if r_idx is not None:
        r_idx = _idx(r_idx, "r_idx").contiguous()
        if r_idx.shape != h_idx.shape:
            raise ValueError("r_idx must match h_idx")
    B, G = t_idx.shape
    if j_major_out:
        out = torch.empty((G, B), dtype=torch.float32, device=h_idx.device).t()
    else:
        out = torch.empty((B, G), dtype=torch.float32, device=h_idx.device)
    rc = lib.hw_kge_score_grouped(
        C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), _ptr(t_idx), t_idx.stride(0), t_idx.stride(1), B, G,
        _ptr(out), out.stride(0), out.stride(1), _ptr(_OobCounter.get(h_idx.device)), _stream(),
    )
    check(rc, "hw_kge_score_grouped")
    return out


def kge_score_grouped_sampled_raw(desc, h_idx, r_idx, r_fixed, t_idx, used_ptr, used_idx, keys, value_num, seed, offset,
                                 out=None):
    """Scores [B, 1 + K] with the K negatives of every group drawn inside the scorer (column 0 of ``t_idx`` is the
    positive; columns 1.. are overwritten with the ids hw_neg_sample would draw for ``keys`` under (seed, offset))."""
    lib = _lib.load()
    h_idx = _idx(h_idx, "h_idx").contiguous()
    t_idx = _idx(t_idx, "t_idx")
    keys = _idx(keys, "keys").contiguous()
    if t_idx.dim() != 2 or t_idx.shape[0] != h_idx.shape[0] or keys.shape != h_idx.shape:
        raise ValueError("t_idx must be [n_groups, 1 + K]; keys one per group")
    if r_idx is not None:
        r_idx = _idx(r_idx, "r_idx").contiguous()
    B, G = t_idx.shape
    if out is None:
        out = torch.empty((B, G), dtype=torch.float32, device=h_idx.device)
    used_ptr, used_idx = _idx(used_ptr, "used_ptr").contiguous(), _idx(used_idx, "used_idx").contiguous()
    rc = lib.hw_kge_score_grouped_sampled(
        C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), _ptr(t_idx), t_idx.stride(0), t_idx.stride(1), B, G,
        _ptr(used_ptr), _ptr(used_idx), used_ptr.numel() - 1, _ptr(keys), int(value_num), C.c_uint64(seed),
        C.c_uint64(offset), _ptr(out), out.stride(0), out.stride(1), _ptr(_OobCounter.get(h_idx.device)), _stream())
    check(rc, "hw_kge_score_grouped_sampled")
    return out


def kge_score_grouped_sampled_selfadv_raw(desc, h_idx, r_idx, r_fixed, t_idx, used_ptr, used_idx, keys, value_num, seed,
                                         offset, alpha=1.0):
    """kge_score_grouped_sampled_raw + the self-adversarial loss of the scores in the same launch: (scores, loss [1]).
    Raises HopwiseAmdError(unsupported) when a group cannot be owned by one workgroup."""
    lib = _lib.load()
    h_idx = _idx(h_idx, "h_idx").contiguous()
    t_idx = _idx(t_idx, "t_idx")
    keys = _idx(keys, "keys").contiguous()
    if r_idx is not None:
        r_idx = _idx(r_idx, "r_idx").contiguous()
    B, G = t_idx.shape
    dev = h_idx.device
    out = torch.empty((B, G), dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(B, dtype=torch.float32, device=dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    used_ptr, used_idx = _idx(used_ptr, "used_ptr").contiguous(), _idx(used_idx, "used_idx").contiguous()
    rc = lib.hw_kge_score_grouped_sampled_selfadv(
        C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), _ptr(t_idx), t_idx.stride(0), t_idx.stride(1), B, G,
        _ptr(used_ptr), _ptr(used_idx), used_ptr.numel() - 1, _ptr(keys), int(value_num), C.c_uint64(seed),
        C.c_uint64(offset), _ptr(out), out.stride(0), out.stride(1), float(alpha), _ptr(loss), _ptr(ws), _ptr(counter),
        _ptr(_OobCounter.get(dev)), _stream())
    check(rc, "hw_kge_score_grouped_sampled_selfadv")
    return out, loss, counter


class SelfAdvChain:
    """A stream of hw_kge_score_grouped_sampled_selfadv batches without the per-batch finish launch: batch i's group terms
    are added by block 0 of batch i + 1 (hw_kge_score_grouped_sampled_selfadv_chained), the last one by flush().  Every
    loss lands in the [1] tensor given to (or returned by) its launch() -- after the NEXT launch() or flush() on the stream."""

    def __init__(self, n_groups, device):
        self.n = int(n_groups)
        self.ws = [torch.empty(self.n, dtype=torch.float32, device=device) for _ in range(2)]
        self.cur = 0
        self.pending = None  # the loss tensor of the launch whose group terms sit in ws[cur ^ 1]

    def launch(self, desc, h_idx, r_idx, r_fixed, t_idx, used_ptr, used_idx, keys, value_num, seed, offset, alpha=1.0,
               out=None, loss=None):
        lib = _lib.load()
        h_idx = _idx(h_idx, "h_idx").contiguous()
        t_idx = _idx(t_idx, "t_idx")
        keys = _idx(keys, "keys").contiguous()
        if r_idx is not None:
            r_idx = _idx(r_idx, "r_idx").contiguous()
        B, G = t_idx.shape
        if B != self.n:
            raise ValueError(f"chain built for {self.n} groups, batch has {B}")
        dev = h_idx.device
        if out is None:
            out = torch.empty((B, G), dtype=torch.float32, device=dev)
        if loss is None:
            loss = torch.empty(1, dtype=torch.float32, device=dev)
        used_ptr, used_idx = _idx(used_ptr, "used_ptr").contiguous(), _idx(used_idx, "used_idx").contiguous()
        prev = self.pending
        rc = lib.hw_kge_score_grouped_sampled_selfadv_chained(
            C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), _ptr(t_idx), t_idx.stride(0), t_idx.stride(1), B, G,
            _ptr(used_ptr), _ptr(used_idx), used_ptr.numel() - 1, _ptr(keys), int(value_num), C.c_uint64(seed),
            C.c_uint64(offset), _ptr(out), out.stride(0), out.stride(1), float(alpha), _ptr(self.ws[self.cur]),
            _ptr(self.ws[self.cur ^ 1]) if prev is not None else None, self.n if prev is not None else 0, _ptr(prev),
            _ptr(_OobCounter.get(dev)), _stream())
        check(rc, "hw_kge_score_grouped_sampled_selfadv_chained")
        self.pending = loss
        self.cur ^= 1
        return out, loss

    def flush(self):
        if self.pending is not None:
            check(_lib.load().hw_selfadv_finish(_ptr(self.ws[self.cur ^ 1]), self.n, _ptr(self.pending), _stream()),
                  "hw_selfadv_finish")
            self.pending = None


def kge_full_sort_raw(desc, h_idx, r_idx, r_fixed, lo, hi):
    lib = _lib.load()
    h_idx = _idx(h_idx, "h_idx").contiguous()
    if r_idx is not None:
        r_idx = _idx(r_idx, "r_idx").contiguous()
    B = h_idx.shape[0]
    out = torch.empty((B, hi - lo), dtype=torch.float32, device=h_idx.device)
    if B == 0:  # no query: the [0, N] matrix the reference would return (the entry point refuses null buffers)
        return out
    rc = lib.hw_kge_full_sort(
        C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), B, int(lo), int(hi), _ptr(out), hi - lo,
        _ptr(_OobCounter.get(h_idx.device)), _stream(),
    )
    check(rc, "hw_kge_full_sort")
    return out


# ---- gradient sinks: persistent p.grad buffers the gradient kernels accumulate into directly (optim.FusedAdam) ----
_GRAD_SINKS = {}  # parameter data_ptr -> (weakref to the parameter, owner state with a .dirty flag)


def register_grad_sink(param, state):
    import weakref

    _GRAD_SINKS[param.data_ptr()] = (weakref.ref(param), state)


def _live_sink(t):
    """(grad buffer, state) when table ``t`` is a registered parameter whose .grad is in place, else None."""
    hit = _GRAD_SINKS.get(t.data_ptr())
    if hit is None:
        return None
    p = hit[0]()
    if p is None:
        del _GRAD_SINKS[t.data_ptr()]
        return None
    hit[1].check_distributed()
    g = p.grad
    if g is None or g.shape != t.shape or g.dtype != torch.float32 or not g.is_contiguous() or p.shape != t.shape:
        return None
    return g, hit[1]


class KgeScoreGrouped(torch.autograd.Function):
    """Differentiable grouped scorer.  Inputs after the fixed arguments are the head, relation and
    tail tables in slot order; the same Parameter may appear in several slots (kg side: entity tables
    are both head and tail) -- its gradient buffer is then shared and returned once."""

    @staticmethod
    def forward(ctx, model, margin, eps, h_idx, r_idx, r_fixed, t_idx, j_major_out, tail_rows, *tables):
        nh, nr, nt = _N_SLOT[model]
        head, rel, tail = tables[:nh], tables[nh : nh + nr], tables[nh + nr :]
        desc = make_desc(model, head, rel, tail, margin, eps, tail_rows)
        out = kge_score_grouped_raw(desc, h_idx, r_idx, r_fixed, t_idx, j_major_out)
        ctx.model, ctx.margin, ctx.eps, ctx.r_fixed, ctx.tail_rows = model, margin, eps, r_fixed, tail_rows
        ctx.has_r = r_idx is not None
        ctx.save_for_backward(h_idx, t_idx, *([r_idx] if r_idx is not None else []), *tables)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        saved = ctx.saved_tensors
        h_idx, t_idx = saved[0], saved[1]
        off = 2
        r_idx = None
        if ctx.has_r:
            r_idx = saved[2].contiguous()
            off = 3
        tables = saved[off:]
        model = ctx.model
        nh, nr, nt = _N_SLOT[model]
        head, rel, tail = tables[:nh], tables[nh : nh + nr], tables[nh + nr :]
        desc = make_desc(model, head, rel, tail, ctx.margin, ctx.eps, ctx.tail_rows)
        g = g.contiguous()  # [B, G] row-major
        bufs = {}
        grads_out = []
        gs = KgeGrads()
        slots = [("head", nh), ("rel", nr), ("tail", nt)]
        k = 0
        for slot, n in slots:
            arr = getattr(gs, slot)
            for i in range(_lib.MAX_TABLES):
                arr[i] = None
            for i in range(n):
                t = tables[k]
                need = ctx.needs_input_grad[9 + k]
                if need:
                    key = t.data_ptr()
                    if key in bufs:
                        grads_out.append(None)  # shared buffer already returned for an earlier slot
                    elif (sink := _live_sink(t)) is not None and torch.is_grad_enabled() is False:
                        bufs[key] = sink[0]  # accumulate straight into p.grad; autograd gets nothing to add
                        sink[1].touch(t.data_ptr(), sink=True)
                        grads_out.append(None)
                    else:
                        bufs[key] = torch.zeros_like(t)
                        grads_out.append(bufs[key])
                    arr[i] = bufs[key].data_ptr()
                else:
                    grads_out.append(None)
                k += 1
        h_idx = h_idx.contiguous()
        B, G = t_idx.shape
        rel_ws = None
        n_ws = int(lib.hw_kge_grad_grouped_workspace_floats(C.byref(desc), B, G, 1 if r_idx is None else 0))
        if n_ws > 0:
            rel_ws = torch.empty(n_ws, dtype=torch.float32, device=g.device)
        rc = lib.hw_kge_grad_grouped(
            C.byref(desc), C.byref(gs), _ptr(h_idx), _ptr(r_idx), int(ctx.r_fixed), _ptr(t_idx), t_idx.stride(0),
            t_idx.stride(1), B, G, _ptr(g), g.stride(0), g.stride(1), _ptr(rel_ws), _stream(),
        )
        check(rc, "hw_kge_grad_grouped")
        return (None,) * 9 + tuple(grads_out)


_GRAD_WS_FLOATS = {}  # (model, dim, groups, tails, fixed relation) -> hw_kge_grad_grouped_workspace_floats


class PairSide:
    """One side (rec or kg) of a pairwise training loss: positions of its head / relation / tail tables in the parameter
    list handed to :class:`FusedPairLoss`, and its ids (``t_idx`` [B, G], column 0 the positive)."""

    __slots__ = ("head", "rel", "tail", "h_idx", "r_idx", "r_fixed", "t_idx", "tail_rows", "pos_weight")

    def __init__(self, head, rel, tail, h_idx, r_idx, r_fixed, t_idx, tail_rows=None, pos_weight=1.0):
        self.head, self.rel, self.tail = tuple(head), tuple(rel), tuple(tail)
        self.h_idx, self.r_idx, self.r_fixed, self.t_idx = h_idx, r_idx, int(r_fixed), t_idx
        self.tail_rows, self.pos_weight = tail_rows, float(pos_weight)


def _grad_slots(desc_tables, needs, bufs, grads_out_by_pos, positions):
    """Fill a KgeGrads for one launch: every table that needs a gradient gets a buffer -- the parameter's own .grad when
    optim.FusedAdam registered it as a sink, a zero-filled temporary otherwise -- shared between slots and sides that
    use the same parameter (``bufs`` is keyed by storage pointer, ``grads_out_by_pos`` by parameter position)."""
    gs = KgeGrads()
    for slot, tabs, pos in zip(("head", "rel", "tail"), desc_tables, positions):
        arr = getattr(gs, slot)
        for i in range(_lib.MAX_TABLES):
            arr[i] = None
        for i, (t, p) in enumerate(zip(tabs, pos)):
            if not needs[p]:
                continue
            key = t.data_ptr()
            if key not in bufs:
                sink = _live_sink(t) if not torch.is_grad_enabled() else None
                if sink is not None:
                    bufs[key] = sink[0]
                    sink[1].touch(t.data_ptr(), sink=True)
                else:
                    bufs[key] = torch.zeros_like(t)
                    grads_out_by_pos[p] = bufs[key]
            arr[i] = bufs[key].data_ptr()
    return gs


class FusedPairLoss(torch.autograd.Function):
    """calculate_loss of the pairwise gather models as ONE autograd node: both sides' scorers write into the loss
    kernel's input buffer (no cat / stack), the loss kernel leaves dLoss/dscore next to the loss, and the backward is the
    two gradient launches.  The per-op graph (two scorer nodes, views, one or two loss nodes, an add) costs ~0.25 ms of
    host time per step under an eager trainer loop, several times the device time of the step.

    kind: "bce" -- sum over the sides of BCEWithLogits(mean) over [positives; negatives] (rotate.py:118-131,
    complex.py:115-128, analogy.py:196-198 with scale = 2), a side's positives weighted ``pos_weight`` (tiled batches
    scored as groups); "margin" / "margin_sigmoid" -- MarginRankingLoss over the concatenated sides (transe.py:86-96
    with eps, distmult.py:86-93, rescal.py:73-99, hole.py:100-104), pairs only."""

    @staticmethod
    def forward(ctx, model, kind, margin, eps, scale, sides, *params):
        lib = _lib.load()
        dev = params[0].device
        st = _stream()
        oob = _ptr(_OobCounter.get(dev))
        ctx.model, ctx.kind, ctx.margin, ctx.eps, ctx.scale, ctx.sides = model, kind, margin, eps, scale, sides
        need_g = any(ctx.needs_input_grad[6:])
        idx = []
        for s in sides:
            h = _idx(s.h_idx, "h_idx").contiguous()
            r = None if s.r_idx is None else _idx(s.r_idx, "r_idx").contiguous()
            t = _idx(s.t_idx, "t_idx")
            if t.dim() != 2 or t.shape[0] != h.shape[0]:
                raise ValueError("t_idx must be [n_groups, n_tail_per_group]")
            idx.append((h, r, t))
        descs = [make_desc(model, [params[i] for i in s.head], [params[i] for i in s.rel], [params[i] for i in s.tail],
                           margin, eps, s.tail_rows) for s in sides]
        ws = _workspace(dev)
        gbufs = []
        if kind == "bce":
            losses = torch.empty(len(sides), dtype=torch.float32, device=dev)
            sizes = [t.shape[0] * t.shape[1] for _, _, t in idx]
            # one buffer for every side's scores and one for their gradients (a single scaling launch in the backward)
            buf_all = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            g_all = torch.empty_like(buf_all) if need_g else None
            off = 0
            for k, (s, d, (h, r, t)) in enumerate(zip(sides, descs, idx)):
                B, G = t.shape
                bptr = buf_all.data_ptr() + 4 * off  # [G, B] j-major: the B positives first
                check(lib.hw_kge_score_grouped(C.byref(d), _ptr(h), _ptr(r), s.r_fixed, _ptr(t), t.stride(0), t.stride(1),
                                               B, G, C.c_void_p(bptr), 1, B, oob, st), "hw_kge_score_grouped")
                gptr = C.c_void_p(g_all.data_ptr() + 4 * off) if need_g else None
                check(lib.hw_loss_bce_logits_weighted(C.c_void_p(bptr), G * B, B, s.pos_weight,
                                                      C.c_void_p(losses.data_ptr() + 4 * k), gptr, _ptr(ws), st),
                      "hw_loss_bce_logits_weighted")
                off += sizes[k]
            gbufs.append(g_all)
            loss = losses.sum() if len(sides) > 1 else losses[0]
        else:
            if any(t.shape[1] != 2 for _, _, t in idx):
                raise ValueError("margin losses take (positive, negative) pairs")
            n_tot = sum(t.shape[0] for _, _, t in idx)
            buf = torch.empty((2, n_tot), dtype=torch.float32, device=dev)
            off = 0
            for s, d, (h, r, t) in zip(sides, descs, idx):
                B = t.shape[0]
                check(lib.hw_kge_score_grouped(C.byref(d), _ptr(h), _ptr(r), s.r_fixed, _ptr(t), t.stride(0), t.stride(1),
                                               B, 2, C.c_void_p(buf.data_ptr() + 4 * off), 1, n_tot, oob, st),
                      "hw_kge_score_grouped")
                off += B
            g = torch.empty_like(buf) if need_g else None
            losses = torch.empty(1, dtype=torch.float32, device=dev)
            fn = lib.hw_loss_margin_ranking_sigmoid if kind == "margin_sigmoid" else lib.hw_loss_margin_ranking
            check(fn(_ptr(buf[0]), _ptr(buf[1]), n_tot, float(margin), _ptr(losses), _ptr(g[0]) if need_g else None,
                     _ptr(g[1]) if need_g else None, _ptr(ws), st), "hw_loss_margin_ranking")
            gbufs.append(g)
            loss = losses[0]
        ctx.n_idx = [(h, r, t) for h, r, t in idx]
        ctx.descs = descs  # raw table pointers: valid as long as the saved parameters are
        ctx.save_for_backward(*params, *[x for x in gbufs if x is not None])
        ctx.n_params = len(params)
        if scale != 1.0:
            loss = loss * scale
        return loss

    @staticmethod
    def backward(ctx, g_up):
        lib = _lib.load()
        saved = ctx.saved_tensors
        params, gbufs = saved[: ctx.n_params], saved[ctx.n_params :]
        needs = ctx.needs_input_grad[6:]
        st = _stream()
        coef = g_up * ctx.scale if ctx.scale != 1.0 else g_up
        bufs, out_by_pos = {}, {}
        off = 0
        n_tot = gbufs[0].shape[1] if ctx.kind != "bce" else 0
        gall = gbufs[0] * coef  # every side's dLoss/dscore, scaled by the upstream gradient in one launch
        for k, (s, (h, r, t)) in enumerate(zip(ctx.sides, ctx.n_idx)):
            head, rel, tail = [params[i] for i in s.head], [params[i] for i in s.rel], [params[i] for i in s.tail]
            desc = ctx.descs[k]
            gs = _grad_slots((head, rel, tail), needs, bufs, out_by_pos, (s.head, s.rel, s.tail))
            B, G = t.shape
            if ctx.kind == "bce":
                gptr, gsb, gsj = gall.data_ptr() + 4 * off, 1, B  # this side's [G, B] block
                off += B * G
            else:
                gptr, gsb, gsj = gall.data_ptr() + 4 * off, 1, n_tot  # columns of the shared [2, n_tot] matrix
                off += B
            rel_ws = None
            wkey = (ctx.model, desc.dim, B, G, r is None)
            n_ws = _GRAD_WS_FLOATS.get(wkey)
            if n_ws is None:
                n_ws = _GRAD_WS_FLOATS[wkey] = int(
                    lib.hw_kge_grad_grouped_workspace_floats(C.byref(desc), B, G, 1 if r is None else 0))
            if n_ws > 0:
                rel_ws = torch.empty(n_ws, dtype=torch.float32, device=t.device)
            check(lib.hw_kge_grad_grouped(C.byref(desc), C.byref(gs), _ptr(h), _ptr(r), s.r_fixed, _ptr(t), t.stride(0),
                                          t.stride(1), B, G, C.c_void_p(gptr), gsb, gsj, _ptr(rel_ws), st),
                  "hw_kge_grad_grouped")
        return (None,) * 6 + tuple(out_by_pos.get(p) for p in range(ctx.n_params))


def fused_pair_loss(model, kind, margin, eps, scale, sides, params):
    """One autograd node for calculate_loss: through torch.ops.hopwise_amd.pair_loss(_backward) when the registered-op
    library is built (two Python -> C++ crossings per step), else through the ctypes binding (FusedPairLoss)."""
    from . import torch_ops

    if torch_ops.enabled():
        return torch_ops.pair_loss(model, kind, margin, eps, scale, sides, params)
    return FusedPairLoss.apply(model, kind, float(margin), float(eps), float(scale), sides, *params)


def kge_score_grouped(model, head, rel, tail, h_idx, r_idx, r_fixed, t_idx, margin=0.0, eps=0.0, j_major_out=False,
                      tail_rows=None):
    if h_idx.numel() == 0:  # an empty batch scores to an empty matrix, as the reference's gathers would
        nt = t_idx.shape[1] if t_idx.dim() == 2 else 1
        return head[0].new_zeros((nt, 0) if j_major_out else (0, nt))
    return KgeScoreGrouped.apply(model, float(margin), float(eps), h_idx, r_idx, int(r_fixed), t_idx, j_major_out,
                                 tail_rows, *head, *rel, *tail)


def kge_full_sort(model, head, rel, tail, h_idx, r_idx, r_fixed, lo, hi, margin=0.0):
    desc = make_desc(model, head, rel, tail, margin, 0.0)
    return kge_full_sort_raw(desc, h_idx, r_idx, r_fixed, lo, hi)


def kge_full_sort_topk(model, head, rel, tail, h_idx, r_idx, r_fixed, lo, hi, k, history_index=None, mask_first_col=True,
                       margin=0.0, run_if=None, out=None):
    """Masked top-k of the full sort without the [B, N] matrix: (values [B, k], column indices [B, k]).
    ``run_if`` (int32 [1] on the device) + ``out`` = (values, indices): the launches are predicated on ``run_if[0] != 0`` and
    leave ``out`` untouched otherwise (hw_kge_full_sort_topk_if: the exact fallback behind :func:`topk_rerank`)."""
    lib = _lib.load()
    desc = make_desc(model, head, rel, tail, margin, 0.0)
    h_idx = _idx(h_idx, "h_idx").contiguous()
    if r_idx is not None:
        r_idx = _idx(r_idx, "r_idx").contiguous()
    B = h_idx.shape[0]
    if B == 0:
        return (torch.empty((0, k), dtype=torch.float32, device=h_idx.device),
                torch.empty((0, k), dtype=torch.int64, device=h_idx.device))
    hr = hc = None
    nh = 0
    if history_index is not None:
        hr, hc = _idx(history_index[0], "history rows").contiguous(), _idx(history_index[1], "history cols").contiguous()
        nh = hr.numel()
    nbytes = int(lib.hw_kge_full_sort_topk_workspace_bytes(B, hi - lo, int(k)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h_idx.device)
    if out is None:
        val = torch.empty((B, k), dtype=torch.float32, device=h_idx.device)
        idx = torch.empty((B, k), dtype=torch.int64, device=h_idx.device)
    else:
        val, idx = _dev(out[0], torch.float32, "out values"), _dev(out[1], torch.int64, "out indices")
        if val.shape != (B, k) or idx.shape != (B, k) or not val.is_contiguous() or not idx.is_contiguous():
            raise ValueError("out must be contiguous ([B, k] float32, [B, k] int64)")
    if run_if is None:
        check(lib.hw_kge_full_sort_topk(C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), B, int(lo), int(hi), _ptr(hr),
                                        _ptr(hc), nh, int(bool(mask_first_col)), int(k), _ptr(ws), nbytes, _ptr(val),
                                        _ptr(idx), _stream()), "hw_kge_full_sort_topk")
    else:
        run_if = _dev(run_if, torch.int32, "run_if")
        check(lib.hw_kge_full_sort_topk_if(C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), B, int(lo), int(hi),
                                           _ptr(hr), _ptr(hc), nh, int(bool(mask_first_col)), int(k), _ptr(ws), nbytes,
                                           _ptr(val), _ptr(idx), _ptr(run_if), _stream()), "hw_kge_full_sort_topk_if")
    return val, idx


# ---- full sorts as one dense product (csrc/kge_fullsort_gemm.hip) -----------------------------------------------------
GEMM_QUERY_MODELS = ("TransE", "DistMult", "RotatE", "ComplEx", "Analogy")
DIST_RESCORE_FRAC = 1.0 / 16.0  # d^2 below this fraction of ||q||^2 + ||t||^2: recomputed from the rows (cancellation)


def kge_query_rows(model, head, rel, tail, h_idx, r_idx, r_fixed, sqnorm=False, margin=0.0):
    """(query rows [B, NQ * dim], ||query||^2 [B] or None): the folded queries of a full sort as a dense matrix."""
    lib = _lib.load()
    desc = make_desc(model, head, rel, tail, margin, 0.0)
    h_idx = _idx(h_idx, "h_idx").contiguous()
    if r_idx is not None:
        r_idx = _idx(r_idx, "r_idx").contiguous()
    B = h_idx.shape[0]
    K = len(tail) * desc.dim
    out = torch.empty((B, K), dtype=torch.float32, device=h_idx.device)
    qn = torch.empty(B, dtype=torch.float32, device=h_idx.device) if sqnorm else None
    if B:
        check(lib.hw_kge_query_rows(C.byref(desc), _ptr(h_idx), _ptr(r_idx), int(r_fixed), B, _ptr(out), K, _ptr(qn),
                                    _ptr(_OobCounter.get(h_idx.device)), _stream()), "hw_kge_query_rows")
    return out, qn


def rows_sqnorm(x):
    """sum_k x[j, k]^2 per row of a [rows, k] fp32 matrix (unit inner stride)."""
    lib = _lib.load()
    x = _dev(x, torch.float32, "x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be [rows, k] with unit inner stride")
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    if x.shape[0]:
        check(lib.hw_rows_sqnorm(_ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(out), _stream()), "hw_rows_sqnorm")
    return out


def kge_dist_finish_(dots, q, t, qn, tn, base=0.0, rescore_frac=DIST_RESCORE_FRAC, mask_column0=False, counter=None):
    """In place: dots[i, j] = <q_i, t_j>  ->  base - ||q_i - t_j||  (hw_kge_dist_finish)."""
    lib = _lib.load()
    c = _dev(dots, torch.float32, "dots")
    q, t = _dev(q, torch.float32, "q"), _dev(t, torch.float32, "t")
    qn, tn = _dev(qn, torch.float32, "qn").contiguous(), _dev(tn, torch.float32, "tn").contiguous()
    if c.dim() != 2 or c.stride(1) != 1 or q.stride(1) != 1 or t.stride(1) != 1:
        raise ValueError("matrices need unit inner stride")
    m, n = c.shape
    if q.shape[0] != m or t.shape[0] != n or q.shape[1] != t.shape[1] or qn.numel() != m or tn.numel() != n:
        raise ValueError("shape mismatch")
    if m and n:
        check(lib.hw_kge_dist_finish(_ptr(c), c.stride(0), m, n, _ptr(q), q.stride(0), _ptr(t), t.stride(0), q.shape[1],
                                     _ptr(qn), _ptr(tn), float(base), float(rescore_frac), int(bool(mask_column0)),
                                     _ptr(counter), _stream()), "hw_kge_dist_finish")
    return dots


RANK_PRODUCT_REL_ERR = 1.2e-5  # hw_gemm_rank: |error| <= this x sum_k |q_k t_k| (three split terms of six)


def mask_bits(history_index, n_rows, n_cols, column0=True, device=None):
    """What ``mask_scores_`` would set to -inf -- column 0 (the [PAD] item) and the evaluation's history pairs -- as the bit
    matrix ``gemm_rank`` masks with: int32 [n_rows, words] (``hw_mask_bits``)."""
    lib = _lib.load()
    hr = hc = None
    n_hist = 0
    if history_index is not None:
        hr, hc = (_idx(x, "history_index").contiguous() for x in history_index)
        n_hist = hr.numel()
        device = hr.device
    words = int(lib.hw_mask_bits_words(int(n_cols)))
    bits = torch.empty((int(n_rows), words), dtype=torch.int32, device=device)
    check(lib.hw_mask_bits(_ptr(hr), _ptr(hc), n_hist, int(n_rows), int(n_cols), int(bool(column0)), _ptr(bits), words,
                           _stream()), "hw_mask_bits")
    return bits


def gemm_rank(q, t, bias=None, out=None, mask=None, tile_max=None, exact=False):
    """out[M, N] = q @ t^T + bias from the three largest split terms (``hw_gemm_rank``): a ranking product for callers that
    re-score their candidates exactly.  ``mask`` (``mask_bits``): those entries leave as -inf; ``tile_max`` [M, ceil(N / 64)]
    receives every 64-column stretch's maximum (for ``topk_from_tile_max``).  Returns None when the split loop does not apply
    to the operands (the caller takes its six-term route).  ``exact``: all six terms (``hw_gemm_f32_masked``: ``gemm``'s values
    with the same by-products)."""
    lib = _lib.load()
    q = _unit_rows(_dev(q, torch.float32, "q"))
    t = _unit_rows(_dev(t, torch.float32, "t"))
    M, K = q.shape
    N = t.shape[0]
    if out is None:
        out = padded_rows(M, N, q.device)
    rc = (lib.hw_gemm_f32_masked if exact else lib.hw_gemm_rank)(
        _ptr(q), q.stride(0), _ptr(t), t.stride(0), _ptr(bias), _ptr(out), out.stride(0), M, N, K,
                          _ptr(mask), 0 if mask is None else mask.stride(0), _ptr(tile_max),
                          0 if tile_max is None else tile_max.stride(0), _stream())
    if rc == -2:  # HW_ERR_UNSUPPORTED: operands the split loop does not take, or HW_GEMM_SPLIT=0
        return None
    check(rc, "hw_gemm_rank")
    return out


def gemm_lse(q, t, bias=None, out=None, row_lse=None):
    """(out, row_lse): out[M, N] = q @ t^T + bias with ``gemm``'s values and row_lse[M, 2] = (row maximum, log of the row's sum
    of exp(x - maximum)) -- ``F.log_softmax(out, 1)[r, j] == (out[r, j] - row_lse[r, 0]) - row_lse[r, 1]`` to fp32 rounding,
    from the product's epilogue (``hw_gemm_f32_lse`` + ``hw_lse_rows``) instead of a pass over the matrix.  Returns None when the
    split loop does not apply to the operands (the caller takes ``gemm`` + a row pass)."""
    lib = _lib.load()
    q = _unit_rows(_dev(q, torch.float32, "q"))
    t = _unit_rows(_dev(t, torch.float32, "t"))
    M, K = q.shape
    N = t.shape[0]
    if out is None:
        out = padded_rows(M, N, q.device)
    n_st = (N + 63) // 64
    tiles = torch.empty((2, max(M, 1), n_st), dtype=torch.float32, device=q.device)
    rc = lib.hw_gemm_f32_lse(_ptr(q), q.stride(0), _ptr(t), t.stride(0), _ptr(bias), _ptr(out), out.stride(0), M, N, K,
                             _ptr(tiles[0]), _ptr(tiles[1]), n_st, _stream())
    if rc == -2:  # HW_ERR_UNSUPPORTED: operands the split loop does not take, or HW_GEMM_SPLIT=0
        return None
    check(rc, "hw_gemm_f32_lse")
    if row_lse is None:
        row_lse = torch.empty((max(M, 1), 2), dtype=torch.float32, device=q.device)
    check(lib.hw_lse_rows(_ptr(tiles[0]), _ptr(tiles[1]), n_st, M, n_st, _ptr(row_lse), _stream()), "hw_lse_rows")
    return out, row_lse


def rank_planes(x):
    """The (hi, mid) bf16 plane images of an fp32 matrix [rows, K] for ``gemm_rank_planes`` (``hw_rank_planes``): uint8 tensor."""
    lib = _lib.load()
    x = _unit_rows(_dev(x, torch.float32, "x"))
    rows, K = x.shape
    planes = torch.empty(int(lib.hw_rank_planes_bytes(rows, K)), dtype=torch.uint8, device=x.device)
    check(lib.hw_rank_planes(_ptr(x), x.stride(0), rows, K, _ptr(planes), _stream()), "hw_rank_planes")
    return planes


def gemm_rank_planes(q_planes, t_planes, m, n, k, bias=None, out=None, mask=None, tile_max=None):
    """``gemm_rank(q, t, ...)`` from the operands' plane images (``rank_planes``): same values, no conversion in the product."""
    lib = _lib.load()
    if out is None:
        out = padded_rows(m, n, q_planes.device)
    check(lib.hw_gemm_rank_planes(_ptr(q_planes), _ptr(t_planes), int(m), int(n), int(k), _ptr(bias), _ptr(out), out.stride(0),
                                  _ptr(mask), 0 if mask is None else mask.stride(0), _ptr(tile_max),
                                  0 if tile_max is None else tile_max.stride(0), _stream()), "hw_gemm_rank_planes")
    return out


def topk_from_tile_max(scores, tile_max, k, overflow=None):
    """(values [R, k], columns [R, k], overflow int32 [1]) = topk_rows(scores, k) answered from the 64-column stretch maxima
    ``gemm_rank`` left (``hw_topk_from_tile_max``); ``overflow`` != 0: some row could not be answered (the caller falls back)."""
    lib = _lib.load()
    scores = _unit_rows(_dev(scores, torch.float32, "scores"))
    R, N = scores.shape
    val = torch.empty((R, k), dtype=torch.float32, device=scores.device)
    idx = torch.empty((R, k), dtype=torch.int64, device=scores.device)
    if overflow is None:
        overflow = torch.zeros(1, dtype=torch.int32, device=scores.device)
    check(lib.hw_topk_from_tile_max(_ptr(scores), scores.stride(0), R, N, _ptr(tile_max), tile_max.stride(0), int(k), _ptr(val),
                                    _ptr(idx), _ptr(overflow), _stream()), "hw_topk_from_tile_max")
    return val, idx, overflow


def gemm_dist(q, t, qn, tn, base=0.0, rescore_frac=DIST_RESCORE_FRAC, out=None):
    """out[i, j] = base - ||q_i - t_j||: the product with the distance epilogue (hw_gemm_dist), one launch."""
    lib = _lib.load()
    q, t = _dev(q, torch.float32, "q"), _dev(t, torch.float32, "t")
    qn, tn = _dev(qn, torch.float32, "qn").contiguous(), _dev(tn, torch.float32, "tn").contiguous()
    if q.dim() != 2 or t.dim() != 2 or q.stride(1) != 1 or t.stride(1) != 1 or q.shape[1] != t.shape[1]:
        raise ValueError("q [m, k] and t [n, k] with unit inner stride")
    m, n = q.shape[0], t.shape[0]
    if qn.numel() != m or tn.numel() != n:
        raise ValueError("qn / tn: one value per row")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=q.device)
    if m and n:
        check(lib.hw_gemm_dist(_ptr(q), q.stride(0), _ptr(t), t.stride(0), _ptr(out), out.stride(0), m, n, q.shape[1],
                               _ptr(qn), _ptr(tn), float(base), float(rescore_frac), _stream()), "hw_gemm_dist")
    return out


def topk_rerank(cand_idx, cand_approx, cand_exact, k, rel_slack=5e-5, abs_slack=1e-6, unverified=None, qn=None, base=0.0,
                tn_max=0.0, dot=False):
    """(values [R, k], indices [R, k], unverified int32 [1]): the exact top-k out of c <= 64 re-scored candidates per row
    (hw_topk_rerank), with the device-side proof that the candidate set sufficed.  ``qn`` [R] (+ ``base``, ``tn_max``):
    ``cand_approx`` are values <q, t> - ||t||^2 / 2 of the distance form's ranking product and the proof is made in the distance
    domain."""
    lib = _lib.load()
    ci = _idx(cand_idx, "cand_idx").contiguous()
    ca = _dev(cand_approx, torch.float32, "cand_approx").contiguous()
    ce = _dev(cand_exact, torch.float32, "cand_exact").contiguous()
    R, c = ci.shape
    if ca.shape != (R, c) or ce.shape != (R, c):
        raise ValueError("candidate arrays must share one [rows, c] shape")
    val = torch.empty((R, k), dtype=torch.float32, device=ci.device)
    idx = torch.empty((R, k), dtype=torch.int64, device=ci.device)
    if unverified is None:
        unverified = torch.zeros(1, dtype=torch.int32, device=ci.device)
    if R:
        if qn is not None:
            qn = _dev(qn, torch.float32, "qn").contiguous()
            if qn.numel() != R:
                raise ValueError("qn: one value per row")
        if dot:  # product scores: the proof in the product's own domain (qn = ||q||^2 per row, tn_max = max ||t||^2)
            check(lib.hw_topk_rerank_dot(_ptr(ci), _ptr(ca), _ptr(ce), R, c, int(k), float(rel_slack), float(abs_slack),
                                         _ptr(qn), float(tn_max), _ptr(val), _ptr(idx), _ptr(unverified), _stream()),
                  "hw_topk_rerank_dot")
            return val, idx, unverified
        check(lib.hw_topk_rerank(_ptr(ci), _ptr(ca), _ptr(ce), R, c, int(k), float(rel_slack), float(abs_slack), _ptr(qn),
                                 float(base), float(tn_max), _ptr(val), _ptr(idx), _ptr(unverified), _stream()), "hw_topk_rerank")
    return val, idx, unverified


class ProjectRows(torch.autograd.Function):
    """out[i] = M[r_i] @ table[e_i] (TransR, transr.py:67-70); M rows are the [n_rel, d * d] proj_mat_e table."""

    @staticmethod
    def forward(ctx, mats, table, r_idx, r_fixed, e_idx):
        lib = _lib.load()
        mats = _dev(mats, torch.float32, "proj_mat_e")
        table = _dev(table, torch.float32, "table")
        d = table.shape[1]
        if mats.dim() != 2 or mats.shape[1] != d * d or not mats.is_contiguous() or not table.is_contiguous():
            raise ValueError("ProjectRows: need contiguous mats [n_rel, d*d] and table [rows, d]")
        e_idx = _idx(e_idx, "e_idx").contiguous()
        if r_idx is not None:
            r_idx = _idx(r_idx, "r_idx").contiguous()
            if r_idx.shape != e_idx.shape:
                raise ValueError("r_idx must match e_idx")
        out = torch.empty((e_idx.numel(), d), dtype=torch.float32, device=table.device)
        check(lib.hw_kge_project_rows(_ptr(mats), mats.shape[0], _ptr(table), table.shape[0], d, _ptr(r_idx), int(r_fixed),
                                      _ptr(e_idx), e_idx.numel(), _ptr(out), _stream()), "hw_kge_project_rows")
        ctx.r_fixed, ctx.has_r = int(r_fixed), r_idx is not None
        ctx.save_for_backward(mats, table, e_idx, *([r_idx] if r_idx is not None else []))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        mats, table, e_idx = ctx.saved_tensors[:3]
        r_idx = ctx.saved_tensors[3] if ctx.has_r else None
        g = g.contiguous()
        g_m = torch.zeros_like(mats) if ctx.needs_input_grad[0] else None
        g_t = torch.zeros_like(table) if ctx.needs_input_grad[1] else None
        check(lib.hw_kge_project_rows_grad(_ptr(mats), mats.shape[0], _ptr(table), table.shape[0], table.shape[1],
                                           _ptr(r_idx), ctx.r_fixed, _ptr(e_idx), e_idx.numel(), _ptr(g), _ptr(g_t),
                                           _ptr(g_m), _stream()), "hw_kge_project_rows_grad")
        return g_m, g_t, None, None, None


def project_rows(mats, table, r_idx, r_fixed, e_idx):
    return ProjectRows.apply(mats, table, r_idx, int(r_fixed), e_idx)


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
_LOSS_WS_FLOATS = None


def _workspace(device, extra=0):
    global _LOSS_WS_FLOATS
    if _LOSS_WS_FLOATS is None:  # a constant of the library: one ctypes call, not one per loss
        _LOSS_WS_FLOATS = int(_lib.load().hw_loss_workspace_floats())
    return torch.empty(_LOSS_WS_FLOATS + int(extra), dtype=torch.float32, device=device)


class BceLogitsMean(torch.autograd.Function):
    """BCEWithLogitsLoss(mean) over a flat score vector whose first ``n_pos`` entries are positives."""

    @staticmethod
    def forward(ctx, x, n_pos, pos_weight=1.0):
        lib = _lib.load()
        ctx.in_shape = x.shape
        x = _dev(x, torch.float32, "scores").contiguous().view(-1)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x) if x.requires_grad or ctx.needs_input_grad[0] else None
        ws = _workspace(x.device)
        check(lib.hw_loss_bce_logits_weighted(_ptr(x), x.numel(), int(n_pos), float(pos_weight), _ptr(loss), _ptr(gx),
                                              _ptr(ws), _stream()), "hw_loss_bce_logits_weighted")
        ctx.save_for_backward(gx)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (gx,) = ctx.saved_tensors
        return (gx * g).view(ctx.in_shape), None, None


class MarginRankingMean(torch.autograd.Function):
    """mean(max(0, -(pos - neg) + margin)); also TransE's triplet loss on scores = -distance."""

    @staticmethod
    def forward(ctx, pos, neg, margin, sigmoid=False):
        lib = _lib.load()
        ctx.shapes = (pos.shape, neg.shape)
        pos = _dev(pos, torch.float32, "pos").contiguous().view(-1)
        neg = _dev(neg, torch.float32, "neg").contiguous().view(-1)
        loss = torch.empty(1, dtype=torch.float32, device=pos.device)
        gp, gn = torch.empty_like(pos), torch.empty_like(neg)
        ws = _workspace(pos.device)
        fn = lib.hw_loss_margin_ranking_sigmoid if sigmoid else lib.hw_loss_margin_ranking
        check(fn(_ptr(pos), _ptr(neg), pos.numel(), float(margin), _ptr(loss), _ptr(gp), _ptr(gn), _ptr(ws), _stream()),
              "hw_loss_margin_ranking")
        ctx.save_for_backward(gp, gn)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        gp, gn = ctx.saved_tensors
        return (gp * g).view(ctx.shapes[0]), (gn * g).view(ctx.shapes[1]), None, None


class SelfAdversarialMean(torch.autograd.Function):
    """Self-adversarial negative-sampling loss on grouped scores [B, 1+K] (column 0 = positive)."""

    @staticmethod
    def forward(ctx, scores, alpha):
        lib = _lib.load()
        s = _dev(scores, torch.float32, "scores").contiguous()
        B, Gc = s.shape
        loss = torch.empty(1, dtype=torch.float32, device=s.device)
        gs = torch.empty_like(s)
        ws = _workspace(s.device, extra=B)
        check(lib.hw_loss_self_adversarial(_ptr(s), B, Gc, float(alpha), _ptr(loss), _ptr(gs), _ptr(ws), _stream()),
              "hw_loss_self_adversarial")
        ctx.save_for_backward(gs)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (gs,) = ctx.saved_tensors
        return gs * g, None


def bce_logits_mean(x, n_pos, pos_weight=1.0):
    """``pos_weight``: each of the first ``n_pos`` scores stands for that many identical positive rows."""
    return BceLogitsMean.apply(x, n_pos, pos_weight)


def margin_ranking_mean(pos, neg, margin, sigmoid=False):
    """``sigmoid=True``: the loss of sigmoid(pos) vs sigmoid(neg) (HolE)."""
    return MarginRankingMean.apply(pos, neg, margin, sigmoid)


def self_adversarial_mean(scores, alpha):
    return SelfAdversarialMean.apply(scores, alpha)


# ------------------------------------------------------------------------------------------------
# evaluation
# ------------------------------------------------------------------------------------------------
def mask_scores_(scores, history_index=None, column0=True):
    """In place: scores[:, 0] = -inf; scores[history_index] = -inf (trainer.py:731-734).  ``column0=False``: history entries
    only (a shard whose column 0 is not entity 0); entries outside the matrix are ignored.  Rows may be padded (any leading
    dimension, unit column stride)."""
    lib = _lib.load()
    s = _dev(scores, torch.float32, "scores")
    if s.dim() != 2 or (s.shape[1] > 1 and s.stride(1) != 1):
        raise ValueError("scores must be [rows, cols] with unit column stride")
    hr = hc = None
    n = 0
    if history_index is not None:
        hr = _idx(history_index[0], "history rows").contiguous()
        hc = _idx(history_index[1], "history cols").contiguous()
        n = hr.numel()
    if s.shape[0] == 0:
        return scores
    check(lib.hw_mask_scores_ld(_ptr(s), s.shape[0], s.shape[1], max(s.stride(0), s.shape[1]), _ptr(hr), _ptr(hc), n,
                                int(bool(column0)), _stream()), "hw_mask_scores_ld")
    return scores


def topk_rows(scores, k, two_stage=True):
    """Row-wise top-k ordered (value desc, index asc): (values [R, k] fp32, indices [R, k] int64).  Long rows (>= 1024 columns,
    k <= 64) go through the two-stage form (hw_topk_rows_ws: each row read once); ``two_stage=False`` keeps the one-block-per-row
    kernel, for the tests that hold the two against each other."""
    lib = _lib.load()
    s = _dev(scores, torch.float32, "scores")
    if s.dim() != 2 or s.stride(1) != 1:
        raise ValueError("scores must be [rows, cols] with unit column stride")
    R, N = s.shape
    val = torch.empty((R, k), dtype=torch.float32, device=s.device)
    idx = torch.empty((R, k), dtype=torch.int64, device=s.device)
    if R == 0:
        return val, idx
    ws_bytes = int(lib.hw_topk_rows_workspace_bytes(R, N, int(k))) if two_stage else 0
    if ws_bytes:
        ws = torch.empty(ws_bytes // 8, dtype=torch.int64, device=s.device)
        check(lib.hw_topk_rows_ws(_ptr(s), R, N, s.stride(0), int(k), _ptr(val), _ptr(idx), _ptr(ws), ws_bytes, _stream()),
              "hw_topk_rows_ws")
    else:
        check(lib.hw_topk_rows(_ptr(s), R, N, s.stride(0), int(k), _ptr(val), _ptr(idx), _stream()), "hw_topk_rows")
    return val, idx


# ------------------------------------------------------------------------------------------------
# sampler
# ------------------------------------------------------------------------------------------------
def neg_sample(used_ptr, used_idx, keys, num, value_num, seed, offset, alias=None, out=None):
    """out[j*len(keys)+i]: j-th negative of keys[i], never in used(keys[i]) (sampler.py:140-183).
    ``out`` (optional int64): either flat contiguous [num*len(keys)] (the reference layout) or a 2-D
    [len(keys), num] view with any strides -- the draws then land directly in a grouped index buffer."""
    lib = _lib.load()
    used_ptr = _idx(used_ptr, "used_ptr").contiguous()
    used_idx = _idx(used_idx, "used_idx").contiguous()
    keys = _idx(keys, "keys").contiguous()
    sk, sd = 1, keys.numel()
    if out is None:
        out = torch.empty(keys.numel() * int(num), dtype=torch.int64, device=keys.device)
    else:
        out = _idx(out, "out")
        if out.dim() == 2:
            if tuple(out.shape) != (keys.numel(), int(num)):
                raise ValueError("2-D out must be [len(keys), num]")
            sk, sd = out.stride(0), out.stride(1)
        elif not out.is_contiguous() or out.numel() != keys.numel() * int(num):
            raise ValueError("out must be contiguous with num*len(keys) elements")
    ap = ak = aa = None
    na = 0
    if alias is not None:
        ap, ak, aa = alias
        ap = _dev(ap, torch.float32, "alias prob").contiguous()
        ak, aa = _idx(ak, "alias keys").contiguous(), _idx(aa, "alias alias").contiguous()
        na = ap.numel()
    check(
        lib.hw_neg_sample(_ptr(used_ptr), _ptr(used_idx), used_ptr.numel() - 1, _ptr(keys), keys.numel(), int(num),
                          int(value_num), _ptr(ap), _ptr(ak), _ptr(aa), na, C.c_uint64(seed & (2**64 - 1)),
                          C.c_uint64(offset & (2**64 - 1)), _ptr(out), sk, sd, _ptr(_OobCounter.get(keys.device)),
                          _stream()),
        "hw_neg_sample",
    )
    return out


# ------------------------------------------------------------------------------------------------
# row exchange primitives (row-sharded tables)
# ------------------------------------------------------------------------------------------------
def gather_rows(table, idx):
    lib = _lib.load()
    table = _dev(table, torch.float32, "table")
    idx = _idx(idx, "idx").contiguous()
    if table.dim() != 2 or not table.is_contiguous():
        raise ValueError("table must be contiguous [rows, dim]")
    out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    check(lib.hw_gather_rows(_ptr(table), table.shape[0], table.shape[1], _ptr(idx), idx.numel(), _ptr(out),
                             _ptr(_OobCounter.get(table.device)), _stream()), "hw_gather_rows")
    return out


def scatter_add_rows_(table, idx, src):
    lib = _lib.load()
    table = _dev(table, torch.float32, "table")
    idx = _idx(idx, "idx").contiguous()
    src = _dev(src, torch.float32, "src").contiguous()
    if table.dim() != 2 or not table.is_contiguous() or src.shape != (idx.numel(), table.shape[1]):
        raise ValueError("shape mismatch")
    check(lib.hw_scatter_add_rows(_ptr(table), table.shape[0], table.shape[1], _ptr(idx), idx.numel(), _ptr(src),
                                  _stream()), "hw_scatter_add_rows")
    return table


def fill_bytes_(t, byte_value):
    """Every byte of the contiguous HIP tensor ``t`` = byte_value, by kernel launches on the current stream: the graph-safe
    stand-in for hipMemsetAsync (include/hopwise_amd.h: hw_fill_bytes)."""
    lib = _lib.load()
    if not (torch.is_tensor(t) and t.is_cuda and t.is_contiguous()):
        raise ValueError("fill_bytes_: a contiguous HIP tensor is required (no CPU path)")
    check(lib.hw_fill_bytes(C.c_void_p(t.data_ptr()), int(byte_value) & 0xff, t.numel() * t.element_size(), _stream()),
          "hw_fill_bytes")
    return t


# ------------------------------------------------------------------------------------------------
# device-side routing of a row-sharded lookup (csrc/shard.hip)
# ------------------------------------------------------------------------------------------------
def shard_plan(ids, n_rows, n_ranks, cap, overflow, n_tables=1):
    """Route ``ids`` (any shape, int64) of a table with ``n_rows`` rows sharded owner = id % n_ranks.
    Returns (inv like ids, meta int32 [P, 1 + cap]): meta[o, 0] = distinct ids asked of owner o, meta[o, 1:] = their local
    rows (ascending) -- ONE message per peer.  inv[i] = owner * (n_tables * cap) + slot: the index into the compact tables
    of a lookup whose ``n_tables`` tables travel packed as [P, n_tables, cap, d] (see shard_gather).  ``overflow`` is a
    persistent int32 [1] device flag the caller inspects later (no host read here)."""
    lib = _lib.load()
    ids = _idx(ids, "ids")
    flat = ids.reshape(-1).contiguous()
    dev = flat.device
    ws = torch.empty(int(lib.hw_shard_plan_workspace_bytes(int(n_rows), int(n_ranks))), dtype=torch.uint8, device=dev)
    inv = torch.empty(flat.numel(), dtype=torch.int64, device=dev)
    meta = torch.empty((n_ranks, 1 + cap), dtype=torch.int32, device=dev)
    check(lib.hw_shard_plan(_ptr(flat), flat.numel(), int(n_rows), int(n_ranks), int(cap), _ptr(ws), _ptr(inv),
                            int(n_tables) * int(cap), C.c_void_p(meta.data_ptr()), 1 + cap, C.c_void_p(meta.data_ptr() + 4),
                            1 + cap, _ptr(_dev(overflow, torch.int32, "overflow")), _ptr(_OobCounter.get(dev)), _stream()),
          "hw_shard_plan")
    return inv.view(ids.shape), meta


def _meta_ptrs(meta, cap):
    meta = _dev(meta, torch.int32, "meta")
    if meta.dim() != 2 or meta.shape[1] != 1 + cap or not meta.is_contiguous():
        raise ValueError("meta must be contiguous int32 [P, 1 + cap]")
    return C.c_void_p(meta.data_ptr() + 4), C.c_void_p(meta.data_ptr()), meta.shape[0]


def shard_gather(shard, meta, cap, out, table=0, n_tables=1):
    """Owner side: out.view(P, n_tables, cap, d)[p, table, j] = shard[meta[p, 1 + j]] for j < meta[p, 0]; other rows stay
    untouched.  ``out``: float32 [P * n_tables * cap, d], one buffer for all tables of the lookup."""
    lib = _lib.load()
    shard = _dev(shard, torch.float32, "shard")
    local, counts, P = _meta_ptrs(meta, cap)
    out = _dev(out, torch.float32, "out")
    if (shard.dim() != 2 or not shard.is_contiguous() or not out.is_contiguous()
            or tuple(out.shape) != (P * n_tables * cap, shard.shape[1]) or not 0 <= table < n_tables):
        raise ValueError("shape mismatch")
    check(lib.hw_shard_gather(_ptr(shard), shard.shape[0], shard.shape[1], local, 1 + cap, counts, 1 + cap, P, int(cap),
                              _ptr(out), n_tables * cap, table * cap, _ptr(_OobCounter.get(shard.device)), _stream()),
          "hw_shard_gather")
    return out


def shard_scatter_add_(grad_shard, meta, cap, src, table=0, n_tables=1):
    lib = _lib.load()
    grad_shard = _dev(grad_shard, torch.float32, "grad_shard")
    src = _dev(src, torch.float32, "src")
    local, counts, P = _meta_ptrs(meta, cap)
    if (grad_shard.dim() != 2 or not grad_shard.is_contiguous() or not src.is_contiguous()
            or tuple(src.shape) != (P * n_tables * cap, grad_shard.shape[1]) or not 0 <= table < n_tables):
        raise ValueError("shape mismatch")
    check(lib.hw_shard_scatter_add(_ptr(grad_shard), grad_shard.shape[0], grad_shard.shape[1], local, 1 + cap, counts,
                                   1 + cap, P, int(cap), _ptr(src), n_tables * cap, table * cap,
                                   _ptr(_OobCounter.get(grad_shard.device)), _stream()), "hw_shard_scatter_add")
    return grad_shard


# ------------------------------------------------------------------------------------------------
# dense fp32 GEMM on the matrix cores
# ------------------------------------------------------------------------------------------------
def gemm(a, b, a_k_contiguous=True, b_k_contiguous=True, bias=None, elu=False, out=None, m_rows=None,
         drop_mask=None, drop_mode=0, seed=0, offset=0):
    """out[M, N] = f(op(a) @ op(b) + bias);  elu: False / True (ELU) / "relu".

    a: [M, K] (a_k_contiguous) or [K, M] (stored transposed); b: [N, K] (b_k_contiguous, nn.Linear layout) or [K, N].
    Rows may be strided (leading dimension = stride(0)); the inner dimension must have unit stride."""
    lib = _lib.load()
    a = _dev(a, torch.float32, "a")
    b = _dev(b, torch.float32, "b")
    if a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError("gemm operands must be 2-D with unit inner stride")
    M, K = (a.shape if a_k_contiguous else (a.shape[1], a.shape[0]))
    N, Kb = (b.shape if b_k_contiguous else (b.shape[1], b.shape[0]))
    if K != Kb:
        raise ValueError(f"inner dimensions differ: {K} vs {Kb}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ws, ws_n = None, 0
    if ((M + 127) // 128) * ((N + 127) // 128) < 256 and K >= 1024:  # long-K, small output: give split-K room
        splits = min(max(768 // (((M + 127) // 128) * ((N + 127) // 128)), 2), K // 128)
        ws_n = splits * M * N
        ws = torch.empty(ws_n, dtype=torch.float32, device=a.device)
    check(
        lib.hw_gemm_f32(_ptr(a), a.stride(0), int(a_k_contiguous), _ptr(b), b.stride(0), int(b_k_contiguous), _ptr(out),
                        out.stride(0), _ptr(m_rows), M, N, K, _ptr(bias), (2 if elu == "relu" else int(bool(elu))), _ptr(drop_mask),
                        int(drop_mode), C.c_uint64(seed), C.c_uint64(offset), _ptr(ws), ws_n, _stream()),
        "hw_gemm_f32",
    )
    return out


def _unit_rows(t):
    """``t`` as a 2-D operand the kernels take: unit inner stride, rows at any 4-byte-aligned pitch (no copy for a padded view)."""
    return t if t.dim() == 2 and t.stride(1) == 1 else t.contiguous()


def padded_rows(m, n, device, dtype=torch.float32):
    """An [m, n] matrix whose rows start on 128-byte boundaries (pitch n rounded up to 32 elements): what the split GEMM and the
    row kernels want of an operand whose width is whatever a vocabulary happens to be (13 805 entities: every second row of
    the contiguous matrix starts off a 16-byte boundary, and the product that reads it falls back to the exact-FMA loop; a
    product that WRITES such a matrix stores 4 bytes at a time).  16-byte rows are what the kernels require; whole cache lines
    per 256-byte store segment are worth another 3-6 % to the products that write [rows, vocabulary] matrices
    (tools/gemm_pitch_probe.py)."""
    return torch.empty((m, (n + 31) & ~31), dtype=dtype, device=device)[:, :n]


class MatMul(torch.autograd.Function):
    """y = x @ w^T (w in nn.Linear layout [N, K]) with both gradients on the same MFMA kernel.  ``padded``: y is a padded-pitch
    view when N is no multiple of 4 (see :func:`padded_rows`; for callers whose consumers take a row pitch)."""

    @staticmethod
    def forward(ctx, x, w, padded=False):
        ctx.save_for_backward(x, w)
        x, w = x.contiguous(), w.contiguous()
        return gemm(x, w, True, True, out=padded_rows(x.shape[0], w.shape[0], x.device) if padded else None)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _unit_rows(gy)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gemm(gy, w.contiguous(), True, False)  # [M, N] @ [N, K]
        if ctx.needs_input_grad[1]:
            gw = gemm(gy, x.contiguous(), False, False)  # gy^T [N, M] @ x [M, K]
        return gx, gw, None


def matmul_nt(x, w, padded=False):
    return MatMul.apply(x, w, padded)


class MatMulNN(torch.autograd.Function):
    """y = x @ w with w stored [K, N] (e.g. the TuckER core viewed [d, d*d])."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return gemm(x.contiguous(), w.contiguous(), True, False)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gemm(gy, w.contiguous(), True, True)  # [M, N] @ ([K, N])^T
        if ctx.needs_input_grad[1]:
            gw = gemm(x.contiguous(), gy, False, False)  # x^T [K, M] @ gy [M, N]
        return gx, gw


def matmul_nn(x, w):
    return MatMulNN.apply(x, w)


class _BatchNorm2dTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum):
        lib = _lib.load()
        x = _dev(x, torch.float32, "x").contiguous()
        N, Cc, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(int(lib.hw_batchnorm2d_workspace_doubles(Cc)), dtype=torch.float64, device=x.device)
        check(lib.hw_batchnorm2d_forward(_ptr(x), N, Cc, H * W, _ptr(weight), _ptr(bias), float(eps), float(momentum),
                                         _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(rstd), _ptr(y), _ptr(ws),
                                         _stream()), "hw_batchnorm2d_forward")
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        N, Cc, H, W = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dg = torch.empty(Cc, dtype=torch.float32, device=x.device) if weight is not None and ctx.needs_input_grad[1] else None
        db = torch.empty(Cc, dtype=torch.float32, device=x.device) if ctx.has_bias and ctx.needs_input_grad[2] else None
        ws = torch.empty(int(lib.hw_batchnorm2d_workspace_doubles(Cc)), dtype=torch.float64, device=x.device)
        check(lib.hw_batchnorm2d_backward(_ptr(x), _ptr(dy), N, Cc, H * W, _ptr(weight), _ptr(mean), _ptr(rstd), _ptr(dx),
                                          _ptr(dg), _ptr(db), _ptr(ws), _stream()), "hw_batchnorm2d_backward")
        return dx, dg, db, None, None, None, None


class BatchNorm2d(torch.nn.BatchNorm2d):
    """nn.BatchNorm2d whose training-mode forward / backward are the two-pass HIP kernels of csrc/batchnorm.hip (same
    parameters, buffers and state_dict keys; eval mode and anything but fp32 HIP NCHW input go to torch)."""

    def forward(self, x):
        if (self.training and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and self.track_running_stats
                and self.momentum is not None):
            if self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            return _BatchNorm2dTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                           self.momentum)
        return super().forward(x)


class _BatchNorm1dTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum):
        lib = _lib.load()
        x = _dev(x, torch.float32, "x").contiguous()
        N, Cc = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(int(lib.hw_batchnorm1d_workspace_doubles(Cc)), dtype=torch.float64, device=x.device)
        check(lib.hw_batchnorm1d_forward(_ptr(x), N, Cc, _ptr(weight), _ptr(bias), float(eps), float(momentum),
                                         _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(rstd), _ptr(y), _ptr(ws), 1,
                                         _stream()), "hw_batchnorm1d_forward")
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        N, Cc = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dg = torch.empty(Cc, dtype=torch.float32, device=x.device) if weight is not None and ctx.needs_input_grad[1] else None
        db = torch.empty(Cc, dtype=torch.float32, device=x.device) if ctx.has_bias and ctx.needs_input_grad[2] else None
        ws = torch.empty(int(lib.hw_batchnorm1d_workspace_doubles(Cc)), dtype=torch.float64, device=x.device)
        check(lib.hw_batchnorm1d_backward(_ptr(x), _ptr(dy), N, Cc, _ptr(weight), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dg),
                                          _ptr(db), _ptr(ws), _stream()), "hw_batchnorm1d_backward")
        return dx, dg, db, None, None, None, None


class BatchNorm1d(torch.nn.BatchNorm1d):
    """nn.BatchNorm1d on the HIP kernels of csrc/batchnorm.hip (same parameters, buffers, state_dict keys): training mode
    with autograd, evaluation under no_grad; anything else (non-fp32, 3-D input, eval with gradients) goes to torch."""

    def forward(self, x):
        ok = x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.track_running_stats and x.shape[0] > 0
        if ok and self.training and self.momentum is not None and x.shape[0] > 1:
            if self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            return _BatchNorm1dTrain.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                           self.momentum)
        if ok and not self.training and not (torch.is_grad_enabled() and (x.requires_grad or (
                self.weight is not None and self.weight.requires_grad))):
            lib = _lib.load()
            x = x.contiguous()
            y = torch.empty_like(x)
            check(lib.hw_batchnorm1d_forward(_ptr(x), x.shape[0], x.shape[1], _ptr(self.weight), _ptr(self.bias),
                                             float(self.eps), 0.0, _ptr(self.running_mean), _ptr(self.running_var), None, None,
                                             _ptr(y), None, 0, _stream()), "hw_batchnorm1d_forward")
            return y
        return super().forward(x)


_TORCH_F_DROPOUT = torch.nn.functional.dropout


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        lib = _lib.load()
        x = _dev(x, torch.float32, "x").contiguous()
        y = torch.empty_like(x)
        check(lib.hw_dropout(_ptr(x), x.numel(), float(p), C.c_uint64(seed), C.c_uint64(offset), _ptr(y), _stream()),
              "hw_dropout")
        ctx.args = (float(p), int(seed), int(offset))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        p, seed, offset = ctx.args
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.hw_dropout(_ptr(dy), dy.numel(), p, C.c_uint64(seed), C.c_uint64(offset), _ptr(dx), _stream()), "hw_dropout")
        return dx, None, None, None


class Dropout(torch.nn.Dropout):
    """nn.Dropout drawn from a device Philox stream keyed by (seed, rank, stream id) with one offset per call
    (distribution-equal to torch's, reproducible per seed).  ``state_dict`` is empty like nn.Dropout's (reference
    checkpoints load strictly); the call counters ride in the owning model's ``other_parameter`` through
    :class:`DropoutCounters`.  ``stream``: an integer, or a string (e.g. the relation name of a CAFE module) hashed to
    one -- never a process-global counter, so the masks of a model do not depend on what was built before it."""

    def __init__(self, p=0.5, seed=2020, stream=0):
        super().__init__(p)
        from .utils import seed_for_rank

        if isinstance(stream, str):
            import zlib

            stream = 1000 + zlib.crc32(stream.encode())
        self._seed = (seed_for_rank(seed) + 0x632BE59BD9B4E019 * (int(stream) + 1)) & 0xFFFFFFFFFFFFFFFF
        from .philox import PhiloxCounter

        self._philox = PhiloxCounter()  # one offset per call; bound to a device word while a step is captured (philox.py)

    @property
    def calls(self):
        return self._philox.calls

    @calls.setter
    def calls(self, v):
        self._philox.calls = int(v)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        # a patched torch.nn.functional.dropout (tests replaying the reference's recorded masks) keeps its say
        if (not (x.is_cuda and x.dtype == torch.float32) or self.p >= 1.0
                or torch.nn.functional.dropout is not _TORCH_F_DROPOUT):
            return super().forward(x)
        return _DropoutFn.apply(x, self.p, self._seed, self._philox.offset(self._philox.next()))


class DropoutCounters:
    """Model mixin: ``_dropout_calls`` = {module path: calls} of every :class:`Dropout` below the model.  Listed in the
    model's ``other_parameter_name`` it travels with the reference trainer's checkpoint (trainer.py:296-304), so a resumed
    run continues every mask stream where it stopped instead of replaying it from call 1."""

    @property
    def _dropout_calls(self):
        return {n: int(m.calls) for n, m in self.named_modules() if isinstance(m, Dropout)}

    @_dropout_calls.setter
    def _dropout_calls(self, counters):
        mods = dict(self.named_modules())
        for n, c in (counters or {}).items():
            if isinstance(mods.get(n), Dropout):
                mods[n].calls = int(c)


class LinearAct(torch.autograd.Function):
    """y = act(x @ w^T + b) (nn.Linear + optional ReLU / ELU) with the backward on the HIP kernels: the activation's
    derivative is recovered from y, the three gradients are two GEMMs and a column sum."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x, w = x.contiguous(), w.contiguous()
        y = gemm(x, w, True, True, bias=b, elu={0: False, 1: True, 2: "relu"}[act])
        ctx.act = act
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import pgpr_ops as P

        x, w, y = ctx.saved_tensors
        gy = gy.contiguous()
        g = P.act_dropout_backward(gy, y, ctx.act) if ctx.act else gy
        gx = gemm(g, w, True, False) if ctx.needs_input_grad[0] else None
        gw = gemm(g, x, False, False) if ctx.needs_input_grad[1] else None
        gb = P.colsum(g) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


_ACT_ARG = {0: False, 1: True, 2: "relu"}


def linear_act(x, w, b, act=0):
    """act: 0 none, 1 ELU, 2 ReLU."""
    if not torch.is_grad_enabled():  # inference: straight to the GEMM (an autograd.Function.apply costs ~8 us of host time,
        return gemm(x.contiguous(), w.contiguous(), True, True, bias=b, elu=_ACT_ARG[int(act)])  # CAFE makes 40 per call)
    return LinearAct.apply(x, w, b, int(act))


class LevelLoss(torch.autograd.Function):
    """(cross entropy of logits vs target, mean sigmoid(logp[neg] - logp[pos])) of one CAFE level (cafe.py:541-553)."""

    @staticmethod
    def forward(ctx, logits, target, pos, neg):
        lib = _lib.load()
        logits = _unit_rows(_dev(logits, torch.float32, "logits"))
        n, v = logits.shape
        target = _idx(target, "target").contiguous()
        pos = None if pos is None else _idx(pos, "pos").contiguous()
        neg = None if neg is None else _idx(neg, "neg").contiguous()
        lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        ws = torch.empty(2 * n, dtype=torch.float32, device=logits.device)
        check(lib.hw_cafe_level_loss(_ptr(logits), logits.stride(0), n, v, _ptr(target), _ptr(pos), _ptr(neg), _ptr(lse),
                                     _ptr(out), _ptr(ws), _stream()), "hw_cafe_level_loss")
        ctx.has_rank = pos is not None
        ctx.save_for_backward(logits, lse, target, *([pos, neg] if pos is not None else []))
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_ce, g_rank):
        lib = _lib.load()
        logits, lse, target = ctx.saved_tensors[:3]
        pos, neg = (ctx.saved_tensors[3], ctx.saved_tensors[4]) if ctx.has_rank else (None, None)
        n, v = logits.shape
        d = padded_rows(n, v, logits.device)  # the product that consumes it (dX = d . E) wants 16-byte row starts
        # the upstream gradients stay on the device: no host read per level, and the launch can be captured
        g_ce = g_ce.to(torch.float32).contiguous()
        g_rank = g_rank.to(torch.float32).contiguous() if ctx.has_rank else None
        check(lib.hw_cafe_level_loss_grad_dev(_ptr(logits), logits.stride(0), n, v, _ptr(lse), _ptr(target), _ptr(pos),
                                              _ptr(neg), _ptr(g_ce), _ptr(g_rank), _ptr(d), d.stride(0), _stream()),
              "hw_cafe_level_loss_grad_dev")
        return d, None, None, None


def level_loss(logits, target, pos=None, neg=None):
    return LevelLoss.apply(logits, target, pos, neg)


class Conv2d1ch(torch.autograd.Function):
    """nn.Conv2d(1, C, (kh, kw)) on an input [B, 1, H, W] (stride 1, no padding) -- csrc/conv.hip."""

    @staticmethod
    def forward(ctx, x, w, bias):
        lib = _lib.load()
        x = _dev(x, torch.float32, "x").contiguous()
        w = _dev(w, torch.float32, "w").contiguous()
        B, one, H, W = x.shape
        C_, one2, kh, kw = w.shape
        if one != 1 or one2 != 1:
            raise ValueError("Conv2d1ch: one input channel only")
        out = torch.empty((B, C_, H - kh + 1, W - kw + 1), dtype=torch.float32, device=x.device)
        check(lib.hw_conv2d_1ch_forward(_ptr(x), _ptr(w), _ptr(bias), B, C_, H, W, kh, kw, _ptr(out), _stream()),
              "hw_conv2d_1ch_forward")
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        B, _, H, W = x.shape
        C_, _, kh, kw = w.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = torch.empty(C_, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        need_w = dw is not None or db is not None
        if need_w and dw is None:
            dw = torch.empty_like(w)
        ws = torch.empty(int(lib.hw_conv2d_1ch_workspace_floats(C_, kh, kw)), dtype=torch.float32, device=x.device) if need_w else None
        check(lib.hw_conv2d_1ch_backward(_ptr(x), _ptr(w), _ptr(gy), B, C_, H, W, kh, kw, _ptr(dx), _ptr(dw), _ptr(db),
                                         _ptr(ws), _stream()), "hw_conv2d_1ch_backward")
        return dx, (dw if ctx.needs_input_grad[1] else None), db


def conv2d_1ch(x, w, bias=None):
    return Conv2d1ch.apply(x, w, bias)


class GatherRows(torch.autograd.Function):
    """table[idx] with a dense scatter-add backward (nn.Embedding forward/backward on the HIP row kernels)."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return gather_rows(table, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gt = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        scatter_add_rows_(gt, idx, g.contiguous())
        return gt, None


def embedding(table, idx):
    return GatherRows.apply(table, idx)


class TuckerContract(torch.autograd.Function):
    """out[b, :] = x[b, :] @ (w[b] * mask[b])  (tucker.py:76-79 with hidden_dropout1 fused)."""

    @staticmethod
    def forward(ctx, x, w, mask):
        lib = _lib.load()
        x = _dev(x, torch.float32, "x").contiguous()
        w = _dev(w, torch.float32, "w").contiguous()
        B, d = x.shape
        if tuple(w.shape) != (B, d, d) or (mask is not None and tuple(mask.shape) != (B, d, d)):
            raise ValueError("w / mask must be [B, d, d]")
        if mask is not None:
            mask = _dev(mask, torch.float32, "mask").contiguous()
        out = torch.empty((B, d), dtype=torch.float32, device=x.device)
        check(lib.hw_tucker_contract(_ptr(x), _ptr(w), _ptr(mask), B, d, _ptr(out), _stream()), "hw_tucker_contract")
        ctx.save_for_backward(x, w, *([mask] if mask is not None else []))
        return out

    @staticmethod
    def backward(ctx, go):
        lib = _lib.load()
        saved = ctx.saved_tensors
        x, w = saved[0], saved[1]
        mask = saved[2] if len(saved) > 2 else None
        B, d = x.shape
        go = go.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        check(lib.hw_tucker_contract_grad(_ptr(x), _ptr(w), _ptr(mask), _ptr(go), B, d, _ptr(gx), _ptr(gw), _stream()),
              "hw_tucker_contract_grad")
        return gx, gw, None


class TuckerContractDropout(torch.autograd.Function):
    """TuckerContract with the dropout mask drawn inside the kernel (Philox), regenerated in the backward."""

    @staticmethod
    def forward(ctx, x, w, p, seed, offset):
        lib = _lib.load()
        x = _dev(x, torch.float32, "x").contiguous()
        w = _dev(w, torch.float32, "w").contiguous()
        B, d = x.shape
        if tuple(w.shape) != (B, d, d):
            raise ValueError("w must be [B, d, d]")
        out = torch.empty((B, d), dtype=torch.float32, device=x.device)
        check(lib.hw_tucker_contract_dropout(_ptr(x), _ptr(w), B, d, float(p), C.c_uint64(seed), C.c_uint64(offset),
                                             _ptr(out), _stream()), "hw_tucker_contract_dropout")
        ctx.args = (float(p), int(seed), int(offset))
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, go):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        p, seed, offset = ctx.args
        B, d = x.shape
        go = go.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        check(lib.hw_tucker_contract_dropout_grad(_ptr(x), _ptr(w), _ptr(go), B, d, p, C.c_uint64(seed),
                                                  C.c_uint64(offset), _ptr(gx), _ptr(gw), _stream()),
              "hw_tucker_contract_dropout_grad")
        return gx, gw, None, None, None


class TuckerContractShared(torch.autograd.Function):
    """out[b, :] = x[b, :] @ (w[index[b]] * m[b]) with a SMALL set of matrices w [n_mats, d, d] instead of the [B, d, d]
    tensor of tucker.py:74-79: mm(r, W) has one distinct row per distinct relation of the batch -- one on the rec side
    (index None: every row uses w[0]), a few dozen on the kg side.  m[b]: explicit mask [B, d, d], or Philox dropout
    (p > 0, same words as TuckerContractDropout), or none."""

    @staticmethod
    def forward(ctx, x, w, index, mask, p, seed, offset):
        lib = _lib.load()
        x = _dev(x, torch.float32, "x").contiguous()
        w = _dev(w, torch.float32, "w").contiguous()
        B, d = x.shape
        if w.dim() != 3 or tuple(w.shape[1:]) != (d, d) or (mask is not None and tuple(mask.shape) != (B, d, d)):
            raise ValueError("w must be [n_mats, d, d], mask [B, d, d]")
        if index is not None:
            index = _idx(index, "index").contiguous()
            if index.numel() != B:
                raise ValueError("index must have one entry per batch row")
        elif w.shape[0] != 1:
            raise ValueError("without an index w must hold one matrix")
        if mask is not None:
            mask = _dev(mask, torch.float32, "mask").contiguous()
        out = torch.empty((B, d), dtype=torch.float32, device=x.device)
        keep = None
        if mask is None and p > 0 and (x.requires_grad or w.requires_grad):
            # the backward reads the forward's keep flags (4 bits per quad of rows and column) instead of drawing the
            # 4 x 10^7 Philox blocks three more times
            keep = torch.empty(int(lib.hw_tucker_keep_words(B, d)), dtype=torch.int32, device=x.device)
            check(lib.hw_tucker_contract_shared_keep(_ptr(x), _ptr(w), _ptr(index), B, d, float(p), C.c_uint64(seed),
                                                     C.c_uint64(offset), _ptr(out), _ptr(keep), _stream()),
                  "hw_tucker_contract_shared_keep")
        else:
            check(lib.hw_tucker_contract_shared(_ptr(x), _ptr(w), _ptr(index), _ptr(mask), B, d, float(p), C.c_uint64(seed),
                                                C.c_uint64(offset), _ptr(out), _stream()), "hw_tucker_contract_shared")
        ctx.args = (float(p), int(seed), int(offset), index is not None, mask is not None, keep is not None)
        ctx.save_for_backward(x, w, *([index] if index is not None else []), *([mask] if mask is not None else []),
                              *([keep] if keep is not None else []))
        return out

    @staticmethod
    def backward(ctx, go):
        lib = _lib.load()
        p, seed, offset, has_index, has_mask, has_keep = ctx.args
        saved = list(ctx.saved_tensors)
        x, w = saved[0], saved[1]
        index = saved[2] if has_index else None
        keep = saved.pop() if has_keep else None
        mask = saved[-1] if has_mask else None
        B, d = x.shape
        go = go.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = ws = order = None
        n_ws = 0
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            if index is None:
                n_ws = int(lib.hw_tucker_contract_shared_workspace_floats(B, d))
                ws = torch.empty(n_ws, dtype=torch.float32, device=x.device)
            else:
                order = torch.argsort(index, stable=True)  # 2048 ids: rows grouped by matrix
        if keep is not None:
            check(lib.hw_tucker_contract_shared_grad_keep(_ptr(x), _ptr(w), _ptr(index), _ptr(order), w.shape[0], _ptr(keep),
                                                          _ptr(go), B, d, p, _ptr(gx), _ptr(gw), _ptr(ws), n_ws, _stream()),
                  "hw_tucker_contract_shared_grad_keep")
            return gx, gw, None, None, None, None, None
        check(lib.hw_tucker_contract_shared_grad(_ptr(x), _ptr(w), _ptr(index), _ptr(order), w.shape[0], _ptr(mask), _ptr(go),
                                                 B, d, p, C.c_uint64(seed), C.c_uint64(offset), _ptr(gx), _ptr(gw), _ptr(ws),
                                                 n_ws, _stream()), "hw_tucker_contract_shared_grad")
        return gx, gw, None, None, None, None, None


def tucker_contract_shared(x, w, index=None, mask=None, p=0.0, seed=0, offset=0):
    return TuckerContractShared.apply(x, w, index, mask, float(p), int(seed), int(offset))


def tucker_contract(x, w, mask=None):
    return TuckerContract.apply(x, w, mask)


def tucker_contract_dropout(x, w, p, seed, offset):
    return TuckerContractDropout.apply(x, w, float(p), int(seed), int(offset))


class SigmoidBceMean(torch.autograd.Function):
    """BCELoss(sigmoid(logits), scale * flag[col] + add), mean over all entries (tucker.py:111-125)."""

    @staticmethod
    def forward(ctx, logits, flag, scale, add):
        lib = _lib.load()
        x = _dev(logits, torch.float32, "logits").contiguous()
        if flag is not None:
            flag = _dev(flag, torch.uint8, "flag").contiguous()
            if flag.numel() != x.shape[1]:
                raise ValueError("flag must have one entry per column")
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        g = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = torch.empty(int(lib.hw_tucker_bce_workspace_floats()), dtype=torch.float32, device=x.device)
        check(lib.hw_tucker_sigmoid_bce(_ptr(x), x.shape[0], x.shape[1], _ptr(flag), float(scale), float(add), _ptr(loss),
                                        None, _ptr(g), _ptr(ws), _stream()), "hw_tucker_sigmoid_bce")
        ctx.save_for_backward(g)
        return loss.view(())

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        return g * gl, None, None, None


def sigmoid_bce_mean(logits, flag, scale, add):
    return SigmoidBceMean.apply(logits, flag, scale, add)


FUSED_SCORE_BCE = True  # False: GEMM, then the separate sigmoid + BCE sweep over the stored scores


class ScoreAllBceMean(torch.autograd.Function):
    """BCELoss(sigmoid(x @ table^T), scale * flag[col] + add), mean over the [B, N] score matrix (tucker.py:83-84,
    111-125), as one unit: the scores live in a buffer whose rows are padded to 32 floats (aligned tile stores and
    16-byte operand loads for the MFMA GEMM whatever N is), the loss kernel overwrites them with dLoss/dscore in place,
    and the backward scales the two small products instead of the [B, N] matrix."""

    @staticmethod
    def forward(ctx, x, table, bias, flag, scale, add):
        lib = _lib.load()
        x, table = x.contiguous(), table.contiguous()
        B, N = x.shape[0], table.shape[0]
        ld = (N + 31) // 32 * 32
        buf = torch.empty((B, ld), dtype=torch.float32, device=x.device)
        if flag is not None:
            flag = _dev(flag, torch.uint8, "flag").contiguous()
            if flag.numel() != N:
                raise ValueError("flag must have one entry per column")
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        need = any(ctx.needs_input_grad[:3])
        K = x.shape[1]
        if FUSED_SCORE_BCE and x.stride(0) % 4 == 0 and table.stride(0) % 4 == 0:
            # scores never written: the GEMM epilogue turns each tile into dLoss/dscore and sums its loss terms
            n_ws = int(lib.hw_gemm_bce_workspace_floats(B, N))
            ws = torch.empty(n_ws, dtype=torch.float32, device=x.device)
            check(lib.hw_gemm_f32_bce(_ptr(x), x.stride(0), _ptr(table), table.stride(0), _ptr(bias), _ptr(buf), ld, B, N, K,
                                      _ptr(flag), float(scale), float(add), _ptr(loss), _ptr(ws), n_ws, _stream()),
                  "hw_gemm_f32_bce")
        else:
            gemm(x, table, True, True, bias=bias, out=buf[:, :N])
            ws = torch.empty(int(lib.hw_tucker_bce_workspace_floats()), dtype=torch.float32, device=x.device)
            check(lib.hw_tucker_sigmoid_bce_ld(_ptr(buf), ld, B, N, _ptr(flag), float(scale), float(add), _ptr(loss), None,
                                               _ptr(buf) if need else None, _ptr(ws), _stream()),
                  "hw_tucker_sigmoid_bce_ld")
        ctx.n = N
        ctx.save_for_backward(x, table, buf if need else None)
        return loss.view(())

    @staticmethod
    def backward(ctx, gl):
        x, table, buf = ctx.saved_tensors
        from . import pgpr_ops as P

        g = buf[:, : ctx.n]  # dLoss/dscore for gl = 1
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(g, table, True, False) * gl  # [B, N] @ [N, d]
        if ctx.needs_input_grad[1]:
            gw = gemm(g, x * gl, False, False)  # g^T [N, B] @ (gl x) [B, d]: the scalar goes onto the SMALL operand
        if ctx.needs_input_grad[2]:
            gb = P.colsum(g) * gl
        return gx, gw, gb, None, None, None


def score_all_bce_mean(x, table, flag, scale, add, bias=None):
    return ScoreAllBceMean.apply(x, table, bias, flag, scale, add)


def sigmoid(x):
    lib = _lib.load()
    x = _dev(x, torch.float32, "x").contiguous()
    y = torch.empty_like(x)
    check(lib.hw_sigmoid(_ptr(x), x.numel(), _ptr(y), _stream()), "hw_sigmoid")
    return y
