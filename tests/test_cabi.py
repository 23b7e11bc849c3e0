"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol that
include/piplib_amd.h declares, its host-only helpers work, and it refuses to run without
a GPU instead of falling back to a CPU path.  (No compute calls here.)"""
import ctypes as C
import os
import re

import pytest

import pipbatch as pb

HDR = os.path.join(pb.ROOT, "include", "piplib_amd.h")


@pytest.fixture(scope="module")
def lib():
    from piplib_amd import build, engine
    build.build(force=False, verbose=False)
    return engine.lib()


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pipamd_\w+)\s*\(", src)))


def test_header_declares_the_three_layers():
    names = declared_functions()
    for must in ("pipamd_engine_create", "pipamd_batch_load", "pipamd_batch_solve", "pipamd_batch_results",
                 "pipamd_solve_tableau", "pipamd_traiter", "pipamd_dense_pivot_bytes", "pipamd_last_solve_ms"):
        assert must in names


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_workspace_and_pivot_bytes_host_only(lib):
    from piplib_amd.engine import BatchDesc
    d = BatchDesc(10000, 127, 0, 64, -1, 1, 128, 0)
    assert lib.pipamd_dense_pivot_bytes(C.byref(d)) == 2 * 64 * 128 * 8
    n = lib.pipamd_batch_workspace_bytes(C.byref(d))
    assert n > 10000 * 64 * 128 * 8          # at least the tableaux themselves
    assert n < 10000 * 4 * (64 + 128) * 128 * 8  # and not absurdly more
    # shapes beyond the engine limits are rejected, not truncated
    bad = BatchDesc(1, 600, 0, 64, -1, 1, 0, 0)
    assert lib.pipamd_batch_workspace_bytes(C.byref(bad)) == 0
    assert b"exceeds" in lib.pipamd_last_error()
    bad2 = BatchDesc(1, 10, 2, 5, 3, 1, 0, 0)  # bigparm must be a parameter column
    assert lib.pipamd_batch_workspace_bytes(C.byref(bad2)) == 0


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.pipamd_engine_create(C.byref(h), 0)
    assert rc != 0 and not h.value
    from piplib_amd import engine
    with pytest.raises(RuntimeError):
        engine.Engine(0)
