"""The drop-in boundary itself (no compute, runs without a GPU): include/jslp_engine.h, the ctypes binding, the product
library and the test library agree on the set of entry points; the product library loads on a GPU-less box, says what it
is, and refuses to create an engine instead of falling back to anything."""
import ctypes
import os
import re
import subprocess

import pytest

from jslpsolver_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "jslp_engine.h")


def declared_in_header():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # comments mention entry points in prose
    return set(re.findall(r"\b(jslp_[a-z0-9_]+)\s*\(", text))


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("jslp_")}


def product_library():
    """built by __graft_entry__.build(); hipcc cross-compiles without a GPU, so its absence is a failure, not a skip"""
    if not os.path.exists(_capi.HIP_LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _capi.HIP_LIB_PATH


def test_header_and_binding_declare_the_same_entry_points():
    assert declared_in_header() == set(_capi.SYMBOLS)


def test_oracle_library_exports_every_declared_symbol(oracle_lib):
    assert exported(oracle_lib.path) == declared_in_header()
    assert oracle_lib.backend == "oracle-c" and oracle_lib.jslp_device_count() == 0


def test_product_library_exports_every_declared_symbol():
    assert exported(product_library()) == declared_in_header()


def test_product_library_loads_and_fails_loudly_without_a_gpu():
    lib = _capi.Library(product_library())  # resolves every symbol of the binding or raises
    assert lib.backend == "hip-gfx950"
    if lib.jslp_device_count() > 0:
        pytest.skip("a GPU is visible: the refusal path is for GPU-less hosts")
    handle = ctypes.c_void_p()
    rc = lib.jslp_engine_create(ctypes.byref(handle), 0, 3, 3, 3, 1e-8)
    assert rc == _capi.JSLP_ERR_DEVICE and not handle.value
    assert "no CPU fallback" in lib.jslp_last_error().decode()
    with pytest.raises(_capi.EngineError):
        from jslpsolver_amd import Solve
        Solve({"optimize": "x", "opType": "max", "constraints": {"c": {"max": 1}}, "variables": {"v": {"x": 1, "c": 1}}})


def test_missing_product_library_is_an_error(tmp_path, monkeypatch):
    monkeypatch.setattr(_capi, "HIP_LIB_PATH", str(tmp_path / "libjslp_hip.so"))
    monkeypatch.setattr(_capi, "_hip", None)
    with pytest.raises(_capi.EngineError, match="no CPU fallback"):
        _capi.load_hip()
