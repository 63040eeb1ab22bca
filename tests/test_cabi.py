"""CPU checks of the boundary: the library loads and exports every symbol include/hopwise_amd.h declares."""

import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hopwise_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hw_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from hopwise_amd import _lib

    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"
    assert b"gfx950" in lib.hw_version()
    assert lib.hw_loss_workspace_floats() >= 1


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    import ctypes as C

    from hopwise_amd import _lib

    lib = _lib.load()
    d = _lib.KgeDesc()
    d.model, d.dim = 1, 0
    assert lib.hw_kge_score_grouped(C.byref(d), None, None, 0, None, 0, 0, 1, 1, None, 0, 0, None, None) == -1
    assert lib.hw_topk_rows(None, 1, 1, 1, 1, None, None, None) == -1
    assert lib.hw_loss_bce_logits(None, 1, 1, None, None, None, None) == -1
    assert lib.hw_neg_sample(None, None, 1, None, 0, 1, 2, None, None, None, 0, 0, 0, None, 1, 1, None, None) == -1


def test_no_cpu_fallback_in_product():
    """The product package must not import the oracle."""
    import pathlib

    for p in pathlib.Path(ROOT, "hopwise_amd").rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p


def test_single_hip_runtime_mapped():
    """torch bundles its own libamdhip64; the C-ABI library must bind to that same instance."""
    import re

    from hopwise_amd import _lib

    _lib.load()
    maps = open("/proc/self/maps").read()
    assert len(set(re.findall(r"/\S*libamdhip64\S*", maps))) == 1


def test_torch_ops_namespace_registered():
    """csrc/torch_bind.cpp: the TORCH_LIBRARY(hopwise_amd) namespace loads (no GPU needed) and carries the ops SURVEY.md
    §8(b) lists, and a CPU tensor is refused rather than computed on."""
    import pytest
    import torch

    from hopwise_amd import torch_ops

    ns = torch_ops.load()
    for name in ("kge_score_grouped", "kge_score_grouped_backward", "kge_full_sort", "kge_full_sort_topk", "kg_neg_sample",
                 "pair_loss", "pair_loss_backward", "pgpr_expand", "pgpr_collect"):
        assert hasattr(ns, name), name
    schema = str(ns.kge_score_grouped.default._schema)
    assert "Tensor[] head" in schema and "Tensor? r_idx" in schema and schema.endswith("-> Tensor")
    with pytest.raises((RuntimeError, NotImplementedError)):  # no CPU kernel is registered: the dispatcher refuses
        ns.kge_full_sort(0, [torch.zeros(4, 8)], [torch.zeros(2, 8)], [torch.zeros(4, 8)], torch.zeros(2, dtype=torch.int64),
                         None, 0, 0, 4, 0.0)
