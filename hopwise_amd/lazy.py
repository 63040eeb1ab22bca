"""A list that is built when somebody reads it.

hopwise's explainable models return ``(scores, paths)`` from ``explain`` with ``paths`` a Python list of
``[user, item, score, path]`` records; ``ExplainableCollector.eval_batch_collect`` (evaluator/collector.py:310-330) keeps it
only when a ``rec.paths`` metric is registered.  Building the records costs the host far more than the device work of the
search that found them, so the models hand back a :class:`LazyList`: a real ``list`` subclass whose content is produced by
``make()`` on the first read, concatenation (the collector's ``previous + value``) or mutation."""


class LazyList(list):
    def __init__(self, make):
        super().__init__()
        self._make = make

    @property
    def pending(self):
        return self._make is not None

    def _fill(self):
        if self._make is not None:
            make, self._make = self._make, None
            list.extend(self, make())
        return self

    def mapped(self, fn):
        """[fn(x) for x in self], unbuilt for as long as self is."""
        if self._make is None:
            return [fn(x) for x in list.__iter__(self)]
        make = self._make
        return LazyList(lambda: [fn(x) for x in make()])

    def __add__(self, other):
        return list(self._fill()) + list(other)

    def __radd__(self, other):
        return list(other) + list(self._fill())

    __hash__ = None


def _filled(name):
    def method(self, *a, **k):
        self._fill()
        for x in a:  # list.__eq__ / extend / ... read another list's storage directly: an unbuilt one would look empty
            if isinstance(x, LazyList):
                x._fill()
        return getattr(list, name)(self, *a, **k)

    method.__name__ = name
    return method


for _n in ("__len__", "__iter__", "__getitem__", "__eq__", "__ne__", "__repr__", "__contains__", "__reversed__", "__iadd__",
           "__mul__", "__setitem__", "__delitem__", "copy", "count", "index", "sort", "append", "extend", "insert", "pop",
           "remove", "reverse"):
    setattr(LazyList, _n, _filled(_n))
