"""Make ``import hopwise`` (the reference at /root/reference) work in the build container.

Used ONLY by ``tests/golden/make_golden.py`` (fixture generation); never imported by tests,
the product package, bench.py or anything that runs on the GPU box.

Six pure-utility packages the reference imports at module load are not installed in this
image (colorama, colorlog, texttable, cachetools, numba, tensorboard).  None of them takes
part in hot-path arithmetic, so inert stand-ins are registered in ``sys.modules`` before
the import.  Nothing is written into /root/reference.
"""

import logging
import sys
import types

sys.dont_write_bytecode = True


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Blank:
    def __getattr__(self, k):
        return ""


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, k):
        return lambda *a, **kw: ""


def _njit(*a, **k):
    if len(a) == 1 and callable(a[0]):
        return a[0]
    return lambda f: f


def install(reference_root="/root/reference"):
    if "colorama" not in sys.modules:
        _mod("colorama", init=lambda *a, **k: None, Fore=_Blank(), Style=_Blank(), Back=_Blank())
    if "colorlog" not in sys.modules:
        _mod("colorlog", ColoredFormatter=lambda *a, **k: logging.Formatter())
    if "texttable" not in sys.modules:
        _mod("texttable", Texttable=_Inert)
    if "cachetools" not in sys.modules:
        _mod("cachetools", LFUCache=dict)
    if "numba" not in sys.modules:
        _mod(
            "numba",
            njit=_njit,
            jit=_njit,
            prange=range,
            typed=types.SimpleNamespace(List=list, Dict=dict),
            types=types.SimpleNamespace(),
        )
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        _mod("tensorboard")
        _mod("torch.utils.tensorboard", SummaryWriter=_Inert)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
