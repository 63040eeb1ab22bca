"""Counters of the device Philox streams (dropout masks, action draws) that can live on the device.

Every random kernel of the library is keyed by ``(seed, offset)``; a model advances ``offset`` by one *call* per use so that no
mask repeats (the reference draws from torch's generator, whose Philox offset advances the same way).  The offset is a host
integer handed to the launch -- which a HIP graph freezes at capture time: a captured training step would replay the same masks
forever.  The kernels therefore also accept an INDIRECT offset (csrc/hw_common.h ``philox_offset``: bit 63 set, bits 0..47 the
address of a uint64 on the device, bits 48..62 a small addend), resolved when the kernel runs.

:class:`PhiloxCounter` is the host side of that: ``calls`` counts the calls made so far (it rides in the checkpoint through the
model's ``other_parameter``: a resumed run continues the stream); ``offset(call, j)`` is the offset of draw ``j`` of call
``call`` -- the plain integer ``stride * call + j``, or, while the counter is bound to a device word (driver.GraphedTrainStep
during its capture), the indirect encoding of the same number relative to the word.  A replay then bumps the word by
``stride * calls_per_step`` as its last node and the host count follows: eager steps and replays draw one and the same stream.
"""

from __future__ import annotations

import torch

INDIRECT = 1 << 63
_ADDEND_LIMIT = 1 << 15


class PhiloxCounter:
    __slots__ = ("calls", "stride", "_ptr", "_base")

    def __init__(self, stride=1, calls=0):
        self.calls = int(calls)
        self.stride = int(stride)
        self._ptr = None   # address of the bound device word (holds stride * _base until the graph bumps it)
        self._base = 0

    def next(self):
        """Start the next call: its index (what ``self._drop_calls += 1`` was)."""
        self.calls += 1
        return self.calls

    def offset(self, call, j=0):
        if not 0 <= j < self.stride:
            raise ValueError("draw index outside the counter's stride")
        if self._ptr is None:
            return self.stride * call + j
        addend = self.stride * (call - self._base) + j
        if not 0 <= addend < _ADDEND_LIMIT:
            raise RuntimeError(f"{call - self._base} Philox calls inside one captured step: more than the indirect offset holds")
        return INDIRECT | (addend << 48) | self._ptr

    # -- capture protocol (driver.GraphedTrainStep) ---------------------------------------------------------------------
    def bind(self, word):
        """``word``: int64 tensor [1] on the device, already holding ``stride * calls``."""
        self._ptr, self._base = int(word.data_ptr()), self.calls
        if self._ptr >> 48:
            raise RuntimeError("device address beyond 48 bits")

    def unbind(self):
        self._ptr, self._base = None, 0

    def __deepcopy__(self, memo):
        return PhiloxCounter(self.stride, self.calls)

    def __reduce__(self):
        return (PhiloxCounter, (self.stride, self.calls))

    def __repr__(self):
        return f"PhiloxCounter(stride={self.stride}, calls={self.calls}{', bound' if self._ptr is not None else ''})"


def counters_of(model):
    """Every PhiloxCounter of ``model`` and its submodules, in a stable order: [(module path + attribute, counter)]."""
    out = []
    for path, mod in model.named_modules():
        for name, v in sorted(vars(mod).items()):
            if isinstance(v, PhiloxCounter):
                out.append((f"{path}.{name}" if path else name, v))
    return out


class DeviceCounters:
    """The device words of a model's counters for one captured step.

    ``share``: another DeviceCounters of the same model -- the two use ONE set of device words (and one record of the host
    counts the words stand for), each with its own per-step increments: the captures of a step that comes in several forms
    (driver.GraphedStagedTrainStep: one capture per metapath of CAFE) then hand the stream on to each other on the device, with
    no host copy between replays of different forms."""

    def __init__(self, model, device, share=None):
        if share is None:
            self.entries = counters_of(model)
            self.words = torch.zeros(max(len(self.entries), 1), dtype=torch.int64, device=device)
            self._sh = {"expect": None}
        else:
            self.entries, self.words, self._sh = share.entries, share.words, share._sh
        n = max(len(self.entries), 1)
        self.inc = torch.zeros(n, dtype=torch.int64, device=device)   # filled after the capture: stride * calls per step
        self.per_step = [0] * len(self.entries)

    @property
    def _expect(self):
        return self._sh["expect"]

    @_expect.setter
    def _expect(self, v):
        self._sh["expect"] = v

    def _host_values(self):
        return [c.stride * c.calls for _, c in self.entries]

    def sync_words(self):
        """Device words := the host counts (before the capture, and before a replay that follows eager steps)."""
        if self.entries:
            self.words[: len(self.entries)].copy_(torch.tensor(self._host_values(), dtype=torch.int64))
        self._expect = [c.calls for _, c in self.entries]

    def begin_capture(self):
        self.sync_words()
        for i, (_, c) in enumerate(self.entries):
            c.bind(self.words[i : i + 1])

    def bump_in_graph(self):
        """Inside the capture, after the step: words += inc (inc is filled by :meth:`end_capture`)."""
        if self.entries:
            self.words.add_(self.inc)

    def end_capture(self):
        """Nothing ran during the capture: the host counts go back to the step's start; remember how far a step moves them."""
        vals = []
        for i, (_, c) in enumerate(self.entries):
            self.per_step[i] = c.calls - c._base
            vals.append(c.stride * self.per_step[i])
            c.calls = c._base
            c.unbind()
        if self.entries:
            self.inc[: len(self.entries)].copy_(torch.tensor(vals, dtype=torch.int64))
        self._expect = [c.calls for _, c in self.entries]

    def before_replay(self):
        if self._expect != [c.calls for _, c in self.entries]:  # eager steps (a ragged batch, an evaluation) moved the counts
            self.sync_words()

    def after_replay(self):
        for i, (_, c) in enumerate(self.entries):
            c.calls += self.per_step[i]
        self._expect = [c.calls for _, c in self.entries]

    def snapshot(self):
        return [c.calls for _, c in self.entries]

    def restore(self, snap):
        for (_, c), v in zip(self.entries, snap):
            c.calls = v
