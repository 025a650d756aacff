"""The C-ABI library loads and exports every symbol include/bdd_mma.h declares — CPU only, no compute."""
import os
import re

from bdd_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="bdd_mma.h", prefix="bddmma_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_every_symbol_of_the_host_header_is_exported_and_bound():
    L = capi.lib()
    syms = declared_symbols("bdd_ilp.h", "bddilp_")
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/bdd_ilp.h but not exported"
        assert s in capi.ILP_SIGNATURES, f"{s} has no ctypes signature in bdd_amd/capi.py"
    assert set(capi.ILP_SIGNATURES) <= set(syms)


def test_every_declared_symbol_is_exported_and_bound():
    L = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 50
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/bdd_mma.h but not exported"
        assert s in capi.SIGNATURES, f"{s} has no ctypes signature in bdd_amd/capi.py"
    assert set(capi.SIGNATURES) <= set(syms)


def test_null_handles_are_rejected():
    L = capi.lib()
    assert L.bddmma_iteration(None, 0.5) == -1
    assert L.bddmma_nr_variables(None) == 0
    assert L.bddmma_last_error(None) is not None


def test_load_rejects_files_that_are_not_checkpoints(tmp_path):
    import ctypes as C
    L = capi.lib()
    h = C.c_void_p()
    p = tmp_path / "x.bin"
    for blob in (b"", b"BDDMMA04", b"BDDMMA04" + b"\xff" * 64, b"BDDMMA03" + b"\0" * 400, b"BDDMMA01" + b"\0" * 400, bytes(range(256)) * 8):
        p.write_bytes(blob)
        assert L.bddmma_load(C.byref(h), 0, str(p).encode()) == -6 and not h.value      # BDDMMA_ERR_IO, before any device work
    assert L.bddmma_load(C.byref(h), 0, str(tmp_path / "missing.bin").encode()) == -6


def test_pybind_module_is_built_and_reports_errors_without_a_gpu():
    from bdd_amd import bdd_solver_py
    assert {"bdd_solver", "bdd_hip_parallel_mma"} <= set(dir(bdd_solver_py))
    import pytest
    with pytest.raises(RuntimeError):
        bdd_solver_py.bdd_solver("{ this is not json", True)
    with pytest.raises(RuntimeError):
        bdd_solver_py.bdd_solver({"relaxation solver": "cuda parallel mma"}, True).solve()   # no input specified
