"""Derived device tensors that are built lazily and then read from more than one HIP stream.

A tensor built on first use is written on whatever stream that first user runs on.  Another stream that finds the cached
object a microsecond later (host time) may launch a reader before the build kernels have finished (device time), and when the
cache entry is replaced the allocator hands the old block back to the building stream while other streams may still be
reading it.  PGPR's pre-split policy planes and first-layer slot tables (rebuilt after every optimizer step), CAFE's transposed
module weights, the side-by-side tables of the bilinear full sort and the out-of-range counter are all of that kind, and
evaluation deals its searches over side streams (model/knowledge_aware_recommender.py: ``_search_and_collect``).

:class:`FencedValue` keeps the build event next to the value: a reader on another stream waits for it (a device-side wait, no
host stall) and marks the tensors as in use on its stream, once per (value, stream).
"""

import torch


def _tensors(value, out):
    if torch.is_tensor(value):
        out.append(value)
    elif isinstance(value, (list, tuple)):
        for v in value:
            _tensors(v, out)
    elif isinstance(value, dict):
        for v in value.values():
            _tensors(v, out)
    return out


class FencedValue:
    """One cached value (a tensor or any nest of lists / tuples / dicts of tensors) with the fence that makes it safe to read
    on a stream other than the one it was built on."""

    __slots__ = ("key", "value", "_event", "_seen", "_cuda")

    def __init__(self):
        self.key = None
        self.value = None
        self._event = None
        self._seen = set()
        self._cuda = []

    def get(self, key, build):
        """The value for ``key``; ``build()`` makes it (on the current stream) when the key has changed."""
        if self.key is None or self.key != key:
            value = build()
            self._cuda = [t for t in _tensors(value, []) if t.is_cuda]
            self._event, self._seen = None, set()
            if self._cuda:
                dev = self._cuda[0].device
                st = torch.cuda.current_stream(dev)
                if not torch.cuda.is_current_stream_capturing():
                    self._event = torch.cuda.Event()
                    self._event.record(st)
                self._seen.add(st.cuda_stream)
            self.value, self.key = value, key
            return value
        if self._cuda:
            dev = self._cuda[0].device
            st = torch.cuda.current_stream(dev)
            if st.cuda_stream not in self._seen:
                if self._event is not None and not torch.cuda.is_current_stream_capturing():
                    st.wait_event(self._event)
                for t in self._cuda:
                    t.record_stream(st)
                self._seen.add(st.cuda_stream)
        return self.value

    def has(self, key):
        """Whether the value for ``key`` is built (no fence taken)."""
        return self.key is not None and self.key == key

    def drop(self):
        self.key = self.value = self._event = None
        self._seen, self._cuda = set(), []

    # a model holding one of these may be deep-copied or pickled (hopwise's trainers do both): events and stream handles
    # do not travel, the value is rebuilt on first use
    def __deepcopy__(self, memo):
        return FencedValue()

    def __reduce__(self):
        return (FencedValue, ())
