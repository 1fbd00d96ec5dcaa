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
    declared = _declared("pem_spgemm.h")
    assert len(declared) >= 28
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
