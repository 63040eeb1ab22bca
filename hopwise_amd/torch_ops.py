"""``torch.ops.hopwise_amd.*``: the C ABI registered with PyTorch's dispatcher (csrc/torch_bind.cpp, TORCH_LIBRARY) plus
the autograd registration of the differentiable ops (``torch.library.register_autograd``).

    from hopwise_amd import torch_ops; torch_ops.load()
    scores = torch.ops.hopwise_amd.kge_score_grouped(MODEL_IDS["RotatE"], [u_re, u_im], [theta], [e_re, e_im],
                                                     users, None, ui_relation, tails, 1.0, 0.0, False, n_items)

Mirrors how the reference's scorers are used (rotate.py:61-69 on gathered rows, :98-131 calculate_loss, :161-220 full
sorts; sampler.py:140-183; pgpr.py:440-552) -- SURVEY.md §8(b).  ``ops.fused_pair_loss`` (the one-node calculate_loss of
the pairwise models) runs through ``torch.ops.hopwise_amd.pair_loss`` / ``pair_loss_backward`` when the library is
loaded: two Python -> C++ crossings per training step instead of ~20 ctypes calls.  The ctypes path (``hopwise_amd._lib``)
stays for the benchmark's raw kernel timing and as the only dependency-free binding of the C ABI.
"""

from __future__ import annotations

import os

import torch

from . import _lib
from .build import TORCH_LIB

_state = {"loaded": False, "autograd": False}

KIND_IDS = {"bce": 0, "margin": 1, "margin_sigmoid": 2}


def available() -> bool:
    return os.path.exists(TORCH_LIB)


def load():
    """Load libhopwise_amd_torch.so (registers the ``hopwise_amd`` namespace).  Raises when it has not been built."""
    if _state["loaded"]:
        return torch.ops.hopwise_amd
    if not available():
        raise _lib.HopwiseAmdError(f"{TORCH_LIB} is missing: run `python -m hopwise_amd.build` (hipcc, gfx950)")
    _lib.load()  # libhopwise_amd.so first: the binding links against it
    torch.ops.load_library(TORCH_LIB)
    _state["loaded"] = True
    _register_autograd()
    return torch.ops.hopwise_amd


def enabled() -> bool:
    """Whether the model classes route through torch.ops (default when built; HW_CTYPES_ONLY=1 keeps the ctypes path)."""
    if os.environ.get("HW_CTYPES_ONLY", "0") == "1" or not available():
        return False
    if not _state["loaded"]:
        load()
    return True


def _register_autograd():
    if _state["autograd"]:
        return

    def setup(ctx, inputs, output):
        model, head, rel, tail, h_idx, r_idx, r_fixed, t_idx, margin, eps, j_major, tail_rows = inputs
        ctx.n = (len(head), len(rel), len(tail))
        ctx.has_r = r_idx is not None
        ctx.meta = (model, r_fixed, margin, eps, tail_rows)
        ctx.need = [bool(t.requires_grad) for t in (*head, *rel, *tail)]
        ctx.save_for_backward(*head, *rel, *tail, h_idx, t_idx, *([r_idx] if r_idx is not None else []))

    def backward(ctx, g):
        nh, nr, nt = ctx.n
        sv = ctx.saved_tensors
        head, rel, tail = sv[:nh], sv[nh : nh + nr], sv[nh + nr : nh + nr + nt]
        h_idx, t_idx = sv[nh + nr + nt], sv[nh + nr + nt + 1]
        r_idx = sv[nh + nr + nt + 2] if ctx.has_r else None
        model, r_fixed, margin, eps, tail_rows = ctx.meta
        outs = torch.ops.hopwise_amd.kge_score_grouped_backward(model, list(head), list(rel), list(tail), h_idx, r_idx,
                                                                r_fixed, t_idx, margin, eps, tail_rows, g,
                                                                [int(x) for x in ctx.need])
        gs = [o if o.numel() else None for o in outs]  # a table shared by two slots is returned once
        return (None, gs[:nh], gs[nh : nh + nr], gs[nh + nr :], None, None, None, None, None, None, None, None)

    torch.library.register_autograd("hopwise_amd::kge_score_grouped", backward, setup_context=setup)
    _state["autograd"] = True


class PairLoss(torch.autograd.Function):
    """ops.FusedPairLoss on the registered ops: forward = ``pair_loss`` (both sides' scorers + the loss kernel), backward =
    ``pair_loss_backward`` (both sides' gradient launches, accumulating into optim.FusedAdam's sinks where registered)."""

    @staticmethod
    def forward(ctx, model, kind, margin, eps, scale, sides, *params):
        from . import ops

        layout, idx, pw = [], [], []
        for s in sides:
            for pos in (s.head, s.rel, s.tail):
                layout.append(len(pos))
                layout.extend(int(p) for p in pos)
            layout += [int(s.r_fixed), 0 if s.r_idx is None else 1, int(s.tail_rows or 0)]
            idx.append(s.h_idx)
            if s.r_idx is not None:
                idx.append(s.r_idx)
            idx.append(s.t_idx)
            pw.append(float(s.pos_weight))
        need = any(ctx.needs_input_grad[6:])
        loss, dscore = torch.ops.hopwise_amd.pair_loss(ops.MODEL_IDS[model], KIND_IDS[kind], margin, eps, scale, list(params),
                                                       layout, idx, pw, need)
        ctx.args = (ops.MODEL_IDS[model], KIND_IDS[kind], margin, eps, scale, layout)
        ctx.n_params, ctx.n_idx = len(params), len(idx)
        ctx.save_for_backward(*params, *idx, dscore)
        return loss

    @staticmethod
    def backward(ctx, g_up):
        from . import ops

        sv = ctx.saved_tensors
        params, idx, dscore = sv[: ctx.n_params], sv[ctx.n_params : ctx.n_params + ctx.n_idx], sv[-1]
        needs = ctx.needs_input_grad[6:]
        sinks = []
        for t, nd in zip(params, needs):
            sink = ops._live_sink(t) if (nd and not torch.is_grad_enabled()) else None
            if sink is not None:
                sink[1].touch(t.data_ptr())
                sinks.append(sink[0])
            else:
                sinks.append(None)
        model, kind, margin, eps, scale, layout = ctx.args
        outs = torch.ops.hopwise_amd.pair_loss_backward(model, kind, margin, eps, scale, list(params), layout, list(idx),
                                                        dscore, g_up, [int(bool(x)) for x in needs], sinks)
        return (None,) * 6 + tuple(o if o.numel() else None for o in outs)


def pair_loss(model, kind, margin, eps, scale, sides, params):
    return PairLoss.apply(model, kind, float(margin), float(eps), float(scale), sides, *params)
