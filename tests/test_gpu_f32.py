"""GPU: fp32 value type.  The reference pins ValueType = double in main (spgemm.cu:728) while its kernels
are templates on it (spgemm.cu:137, 593).  Bar: structure bit-exact and identical to the fp64 run; values bit-exact
against the oracle's float chain (ascending k, one fmaf per product -- `oracle_spgemm_step3_f32` over the tiled layout
and `oracle_csr_spgemm_f32`, the independent serial Gustavson), and within 1e-5 relative of the fp64 result where no
cancellation is involved.  (Parity unpinned at the reference boundary, like the fp64 path: the reference holds no
vectors for either type.)"""
import os

import numpy as np
import pytest

import cachefmt
from matgen import cases
from prune_ref import expected

pytestmark = pytest.mark.gpu
CASES = cases()
C_NAMES = ["c_tile_rowptr", "c_tile_rowidx", "c_tile_colidx", "pairs_offset", "pairs_a", "pairs_b", "c_mask", "c_tile_nnz_ptr",
           "c_rowptr", "c_rowcolidx", "c_vals"]
T_NAMES = ["tile_keys", "tile_nnz_ptr", "masks", "rowptr", "rowcolidx", "vals", "masks_t", "tile_rowptr", "tile_colidx",
           "tile_colptr", "tile_rowidx", "tile_offsets"]


def _pairs(pkg, oracle, ctx, case):
    rows, cols, I, J, V, tr = case
    V32 = V.astype(np.float32)
    Vw = V32.astype(np.float64)                      # the float values, widened exactly: what the oracle computes on
    gA = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V32, False, dtype=np.float32)
    gB = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V32, True, dtype=np.float32) if tr else gA
    oA = oracle.Tiled(rows, cols, I, J, Vw, False)
    oB = oracle.Tiled(rows, cols, I, J, Vw, True) if tr else oA
    return gA, gB, oA, oB, Vw


@pytest.mark.parametrize("name", list(CASES))
def test_f32_conversion_and_three_steps_match_the_float_oracle(pkg, oracle, ctx, name):
    rows, cols, I, J, V, tr = CASES[name]
    gA, gB, oA, oB, Vw = _pairs(pkg, oracle, ctx, CASES[name])
    assert gA.value_bytes == 4 and gA.dtype == np.float32
    for g, o in ((gA, oA), (gB, oB)):
        for arr in T_NAMES:
            got, want = g.array(arr), getattr(o, arr)
            if arr == "vals":
                assert got.dtype == np.float32
                got = got.astype(np.float64)
            assert np.array_equal(got, want), f"{name}: tiled array {arr}"
    plan = pkg.CPlan(ctx, gA, gB)
    plan.spgemm()
    op = oracle.Plan(oA, oB, f32=True)
    want, counts = expected(op, oA, oB)
    info = plan.info()
    assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == counts
    for arr in C_NAMES:
        got = plan.array(arr)
        if arr == "c_vals":
            assert got.dtype == np.float32
            got = got.astype(np.float64)
        assert np.array_equal(got, want[arr]), f"{name}: plan array {arr}"
    rp, ci, v = plan.export_csr()
    assert v.dtype == np.float32
    sa, sb = oracle.Csr(rows, cols, I, J, Vw, False), oracle.Csr(rows, cols, I, J, Vw, tr)
    rp1, ci1, v1 = oracle.csr_spgemm(sa, sb, f32=True).arrays()
    assert np.array_equal(rp, rp1) and np.array_equal(ci, ci1)
    assert np.array_equal(v.astype(np.float64), v1), "values must equal the ascending-k fmaf chain bit for bit"
    r, c, vv = plan.export_coo()
    r0, c0, vv0 = op.export_coo()
    assert np.array_equal(r, r0) and np.array_equal(c, c0) and np.array_equal(vv.astype(np.float64), vv0)


@pytest.mark.parametrize("name", ["rand_300", "powerlaw_600", "rect_70x40_AAt", "blockrows_1600"])
def test_f32_structure_equals_f64_and_values_are_close(pkg, ctx, name):
    rows, cols, I, J, V, tr = CASES[name]
    out = {}
    for dt in (np.float64, np.float32):
        A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, False, dtype=dt)
        B = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True, dtype=dt) if tr else A
        p = pkg.CPlan(ctx, A, B)
        p.spgemm()
        out[dt] = p.export_csr()
    (rp8, ci8, v8), (rp4, ci4, v4) = out[np.float64], out[np.float32]
    assert np.array_equal(rp8, rp4) and np.array_equal(ci8, ci4)
    # float rounding of the inputs (2^-24 each) and of every fmaf: error bound ~ (terms + 2) * 2^-24 * sum |a||b|
    A8 = np.abs(V)
    bound = 64 * 2.0 ** -24 * max(1.0, float(A8.max()) ** 2)
    assert np.max(np.abs(v4.astype(np.float64) - v8)) <= bound


def test_mixed_value_types_and_wrong_export_are_refused(pkg, ctx):
    rows, cols, I, J, V, _ = CASES["rand_300"]
    A8 = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    A4 = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, dtype=np.float32)
    for a, b in ((A8, A4), (A4, A8)):
        with pytest.raises(pkg.PemError) as e:
            pkg.CPlan(ctx, a, b)
        assert e.value.status == -1 and "value type" in str(e.value)
    p = pkg.CPlan(ctx, A4, A4)
    p.spgemm()
    import ctypes as C
    n = C.c_int64()
    nz, nr = p.info()["nnz_c"], rows
    rp, ci, v = np.zeros(nr + 1, np.int32), np.zeros(nz, np.int32), np.zeros(nz, np.float64)
    rc = pkg.lib().pem_c_export_csr(ctx._h, p._h, C.byref(n), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                    v.ctypes.data_as(C.c_void_p))
    assert rc == -1 and b"fp32" in pkg.lib().pem_last_error()
    assert pkg.lib().pem_c_export_csr(ctx._h, p._h, C.byref(n), None, None, None) == 0 and n.value == nz   # size query is type-free


def test_f32_from_csr_and_row_slices(pkg, oracle, ctx):
    rows, cols, I, J, V, _ = CASES["powerlaw_600"]
    V32 = V.astype(np.float32)
    rp, ci, v = oracle.Csr(rows, cols, I, J, V32.astype(np.float64)).arrays()
    a = pkg.Tiled.from_csr(ctx, rows, cols, rp, ci, v.astype(np.float32), dtype=np.float32)
    b = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V32, dtype=np.float32)
    for name in pkg.T_ARRAYS:
        assert np.array_equal(a.array(name), b.array(name)), name
    full = pkg.CPlan(ctx, a, a)
    full.spgemm()
    frp, fci, fv = full.export_csr()
    bounds = pkg.split_tile_rows(ctx, a, a, 3)
    parts = []
    for g in range(3):
        p = pkg.CPlan(ctx, a, a, int(bounds[g]), int(bounds[g + 1]))
        p.spgemm()
        parts.append(p.export_csr())
    assert np.array_equal(np.concatenate([q[1] for q in parts]), fci)
    assert np.array_equal(np.concatenate([q[2] for q in parts]), fv) and fv.dtype == np.float32


def test_f32_cache_round_trip(pkg, oracle, ctx, tmp_path):
    rows, cols, I, J, V, _ = CASES["rand_300"]
    V32 = V.astype(np.float32)
    T = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V32, dtype=np.float32)
    path = str(tmp_path / "f32.pemtile")
    T.save(path)
    o = oracle.Tiled(rows, cols, I, J, V32.astype(np.float64))
    want = cachefmt.cache_bytes(o.rows, o.cols, o.tile_keys, o.tile_nnz_ptr, o.rowcolidx, o.vals.astype(np.float32), value_bytes=4)
    assert open(path, "rb").read() == want                      # 4-byte values, value_bytes = 4 in the header
    L = pkg.Tiled.load(ctx, path)
    assert L.value_bytes == 4
    for name in pkg.T_ARRAYS:
        assert np.array_equal(L.array(name), T.array(name)), name
    assert os.path.getsize(path) < 128 + 64 * 4 + 12 * o.ntiles + 5 * o.nnz + 64 * 4


def test_cli_fp32(pkg, oracle, standins, tmp_path):
    """`pemspgemm ... --fp32`: same surface, float arithmetic; result files hold the float values (widened for '%.17f')."""
    import importlib
    import subprocess
    hostio = importlib.import_module("pem_spgemm_amd.hostio")
    rows, cols, I, J, V = standins.make("scircuit", scale=0.01)
    mtx = str(tmp_path / "mini.mtx")
    standins.write_mtx(mtx, rows, cols, I, J, V)
    env = dict(os.environ, PEM_RESULT_DIR=str(tmp_path), PEM_CSV=str(tmp_path / "r.csv"), PEM_REPEAT="1")
    cdir = tmp_path / "cache"
    cdir.mkdir()
    for attempt in ("cold", "warm"):
        out = subprocess.run([hostio.CLI_PATH, mtx, "1", "--fp32", "--cache", str(cdir)], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr + out.stdout
        assert "value type: fp32" in out.stdout
        Vw = V.astype(np.float32).astype(np.float64)
        a = oracle.Csr(rows, cols, I, J, Vw)
        rp, ci, v = oracle.csr_spgemm(a, a, f32=True).arrays()
        assert int((tmp_path / "SPGEMM_RESULT_NNZ.txt").read_text()) == len(ci)
        c_file = np.loadtxt(tmp_path / "SPGEMM_RESULT_COLS.txt", dtype=np.int64, ndmin=1)
        v_file = np.loadtxt(tmp_path / "SPGEMM_RESULT_VALS.txt", dtype=np.float64, ndmin=1)
        assert np.array_equal(c_file, ci)
        np.testing.assert_allclose(v_file, v, rtol=0, atol=5e-18 + 1e-17)
    assert (cdir / "mini.mtx.A.f32.pemtile").exists() and "1 loaded, 0 rebuilt" in out.stdout
    # an fp64 run next to it keeps its own cache file and its own (different) values
    out = subprocess.run([hostio.CLI_PATH, mtx, "1", "--cache", str(cdir)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and (cdir / "mini.mtx.A.pemtile").exists()
    v64 = np.loadtxt(tmp_path / "SPGEMM_RESULT_VALS.txt", dtype=np.float64, ndmin=1)
    assert not np.array_equal(v64, v_file)
