"""CPU: the C-ABI library loads and exports every symbol include/hylight_mi.h declares
(no compute calls without a GPU)."""
import os
import re

from hylight_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hylight_mi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hlmi_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_bound_and_exported():
    lib = api.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert name in api.SYMBOLS, f"{name} declared in the header but not bound in api.py"
        assert getattr(lib, name) is not None
    assert sorted(api.SYMBOLS) == declared


def test_version_and_error_without_gpu():
    assert "gfx950" in api.version()
    import torch
    if not torch.cuda.is_available():
        # no device here: a compute entry point must fail loudly, not fall back to the CPU
        try:
            api.filter_chunk("/nonexistent.paf", "/tmp/x.paf", 1000, 2, 0.95)
        except api.HlmiError as e:
            assert e.code in (-3, -2)
        else:
            raise AssertionError("expected HlmiError")
