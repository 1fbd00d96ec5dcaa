"""CPU: the C-ABI libraries load and export every symbol the headers declare; without a GPU
the product refuses to run (no CPU fallback)."""
import ctypes as C
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pem_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol(pkg):
    lib = pkg.lib()
    declared = sorted(set(_declared("pem_spgemm.h") + _declared("pem_test.h")))   # (the test hooks live in the same library)
    assert len(declared) >= 28 and "pem_debug_scan_i32" not in _declared("pem_spgemm.h")
    for sym in declared:
        assert hasattr(lib, sym), f"libpemspgemm_hip.so does not export {sym}"
    assert sorted(pkg.ABI_SYMBOLS) == declared
    assert b"gfx950" in lib.pem_version()


def test_host_library_exports_every_declared_symbol(pkg):
    hostio = importlib.import_module("pem_spgemm_amd.hostio")
    lib = hostio.lib()
    declared = _declared("pem_host.h")
    for sym in declared:
        assert hasattr(lib, sym), f"libpemhost.so does not export {sym}"
    assert sorted(hostio.HOST_SYMBOLS) == declared


def test_mgpu_library_exports_and_assembles_row_pointers(pkg):
    """libpemmgpu.so (the C++ tool's --gpus N path over RCCL): every declared symbol is exported, and the host arithmetic
    of the gather -- where each rank's slice lands, row pointers rebased by the entries in front -- equals numpy's."""
    import numpy as np
    hostio = importlib.import_module("pem_spgemm_amd.hostio")
    lib = hostio.mgpu_lib()
    declared = _declared("pem_mgpu.h")
    for sym in declared:
        assert hasattr(lib, sym), f"libpemmgpu.so does not export {sym}"
    assert sorted(hostio.MGPU_SYMBOLS) == declared
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 8):
        lens = [rng.integers(0, 6, int(rng.integers(0, 40))) for _ in range(n)]            # per-row entry counts of every slice
        if n >= 3:
            lens[1] = np.zeros(0, np.int64)                                                 # a rank with no rows at all
        slices = [np.concatenate([[0], np.cumsum(l)]).astype(np.int32) for l in lens]
        row_off, nnz_off, rowptr = hostio.mgpu_assemble_rowptr(slices, [int(s[-1]) for s in slices])
        want = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.int32) if n else np.zeros(1, np.int32)
        assert np.array_equal(rowptr, want)
        assert np.array_equal(row_off, np.concatenate([[0], np.cumsum([len(l) for l in lens])]))
        assert np.array_equal(nnz_off, np.concatenate([[0], np.cumsum([int(l.sum()) for l in lens])]))
    # the re-cut of the row split from measured times: the C++ host's arithmetic against the Python harness'
    mg = importlib.import_module("pem_spgemm_amd.multigpu")
    for nparts in (1, 2, 3, 8):
        mt = int(rng.integers(nparts, 400))
        w = rng.uniform(0.5, 50.0, mt)
        w[rng.integers(0, mt, 3)] = 5000.0                                                  # a few hub rows
        cuts = np.sort(rng.integers(0, mt + 1, nparts - 1))
        b = np.concatenate([[0], cuts, [mt]]).astype(np.int32)
        ms = rng.uniform(0.2, 0.4, nparts)
        for fixed in (0.0, 0.6 * ms.min()):
            assert np.array_equal(hostio.mgpu_recut_bounds(w, b, ms, fixed), mg.recut_bounds(w, b, ms, fixed=fixed)), (nparts, fixed)
    if not __import__("torch").cuda.is_available():
        m = C.c_void_p()
        devs = (C.c_int * 1)(0)
        assert lib.pem_mgpu_create(1, devs, C.byref(m)) == -7       # PEM_E_NODEVICE: no CPU fallback here either


def test_no_gpu_means_loud_failure(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PemError) as e:
        pkg.Context(0)
    assert e.value.status == -7   # PEM_E_NODEVICE: there is no CPU fallback behind the ABI


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under pem-spgemm_amd/ may import, link or load it"""
    pkgdir = os.path.join(ROOT, "pem-spgemm_amd")
    needles = ("oracle_py", "liboracle", "oracle.h", "load_oracle", "oracle/", "import oracle", "loracle")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for n in needles:
                    assert n not in text, f"{os.path.join(dirpath, f)} references the oracle ({n})"
