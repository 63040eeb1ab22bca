"""philox.PhiloxCounter / DeviceCounters on the host: plain offsets, the indirect encoding the kernels resolve
(csrc/hw_common.h: philox_offset), and the bookkeeping of a captured step (no GPU needed: a CPU tensor stands in for the word)."""

import copy
import pickle

import pytest
import torch

from hopwise_amd.philox import INDIRECT, DeviceCounters, PhiloxCounter, counters_of


def _decode(off, memory):
    assert off >> 63
    return memory[off & ((1 << 48) - 1)] + ((off >> 48) & 0x7FFF)


def test_plain_and_indirect_offsets_name_the_same_stream_position():
    c = PhiloxCounter(stride=4, calls=10)
    word = torch.tensor([4 * 10], dtype=torch.int64)
    plain = []
    for _ in range(3):
        call = c.next()
        plain.append([c.offset(call, j) for j in range(4)])
    assert plain[0] == [44, 45, 46, 47] and plain[2][3] == 4 * 13 + 3
    c.calls = 10
    c.bind(word)
    memory = {word.data_ptr(): int(word[0])}
    for want in plain:
        call = c.next()
        got = [_decode(c.offset(call, j), memory) for j in range(4)]
        assert got == want
    assert all(c.offset(c.calls, j) & INDIRECT for j in range(4))
    with pytest.raises(ValueError):
        c.offset(c.calls, 4)
    c.calls += 1 << 14  # 4 * 2^14 > the 15-bit addend
    with pytest.raises(RuntimeError):
        c.offset(c.calls, 0)
    c.unbind()
    assert c.offset(11, 2) == 46


def test_counters_survive_copies_without_their_binding():
    c = PhiloxCounter(stride=2, calls=7)
    c.bind(torch.zeros(1, dtype=torch.int64))
    for d in (copy.deepcopy(c), pickle.loads(pickle.dumps(c))):
        assert (d.stride, d.calls, d._ptr) == (2, 7, None)


def test_device_counters_follow_a_captured_step():
    class Leaf(torch.nn.Module):
        def __init__(self, stride):
            super().__init__()
            self._philox = PhiloxCounter(stride)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = Leaf(1), Leaf(4)
            self._philox = PhiloxCounter(2, calls=5)

    m = Model()
    names = [n for n, _ in counters_of(m)]
    assert names == ["_philox", "a._philox", "b._philox"]
    dc = DeviceCounters(m, "cpu")
    m.a._philox.calls = 3
    dc.begin_capture()
    assert dc.words[:3].tolist() == [10, 3, 0]
    # the "captured" step: two calls of the model's counter, one of a, three of b
    for ctr, n in ((m._philox, 2), (m.a._philox, 1), (m.b._philox, 3)):
        for _ in range(n):
            assert ctr.offset(ctr.next()) >> 63
    dc.bump_in_graph()       # recorded before inc is known ...
    dc.end_capture()         # ... and nothing ran: counts back at the start, inc filled
    assert [c.calls for _, c in dc.entries] == [5, 3, 0] and dc.per_step == [2, 1, 3] and dc.inc[:3].tolist() == [4, 1, 12]
    assert all(c._ptr is None for _, c in dc.entries)
    for _ in range(2):       # two replays: the graph's last node and the host bookkeeping
        dc.before_replay()
        dc.words.add_(dc.inc)
        dc.after_replay()
    assert [c.calls for _, c in dc.entries] == [9, 5, 6] and dc.words[:3].tolist() == [18, 5, 24]
    m.b._philox.next()       # an eager step in between
    dc.before_replay()
    assert dc.words[:3].tolist() == [18, 5, 28]
    snap = dc.snapshot()
    m._philox.next()
    dc.restore(snap)
    assert m._philox.calls == 9
