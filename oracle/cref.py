"""Build + ctypes binding of oracle/kge_ref.c (TEST INFRASTRUCTURE / cpu_baseline ONLY)."""

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "kge_ref.c")
LIB = os.path.join(HERE, "libhw_oracle.so")
MODEL_IDS = {"TransE": 0, "RotatE": 1, "ComplEx": 2, "DistMult": 3}


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"])
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.hw_oracle_threads.restype = C.c_int
        _lib.hw_oracle_score_grouped.restype = C.c_int
    return _lib


def threads():
    return int(load().hw_oracle_threads())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def score_grouped(model, head, rel, tail, h_idx, r_idx, r_fixed, t_idx, margin=0.0, eps=0.0, out=None):
    """head/rel/tail: tuples of float32 [rows, dim] arrays; t_idx [B, G] int64 -> scores [B, G] float32."""
    lib = load()
    f = lambda x: np.ascontiguousarray(x, np.float32)
    head, rel, tail = [f(x) for x in head], [f(x) for x in rel], [f(x) for x in tail]
    h_idx = np.ascontiguousarray(h_idx, np.int64)
    t_idx = np.ascontiguousarray(t_idx, np.int64)
    r_idx = None if r_idx is None else np.ascontiguousarray(r_idx, np.int64)
    B, G = t_idx.shape
    if out is None:
        out = np.empty((B, G), np.float32)
    rc = lib.hw_oracle_score_grouped(
        C.c_int(MODEL_IDS[model]), C.c_int(head[0].shape[1]), _p(head[0]), _p(head[1] if len(head) > 1 else None),
        _p(rel[0]), _p(rel[1] if len(rel) > 1 else None), _p(tail[0]), _p(tail[1] if len(tail) > 1 else None),
        C.c_float(margin), C.c_float(eps), _p(h_idx), _p(r_idx), C.c_int64(r_fixed), _p(t_idx), C.c_int64(B),
        C.c_int64(G), _p(out),
    )
    if rc != 0:
        raise RuntimeError(f"hw_oracle_score_grouped rc={rc}")
    return out
