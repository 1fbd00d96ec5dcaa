"""GPU: the on-disk cache of the tiled format (include/pem_spgemm.h pem_tiled_save / pem_tiled_load).

The file format is restated independently in tests/cachefmt.py.  Parity: a file written from the ORACLE's arrays loads
into a tiling whose twelve reference arrays equal the oracle's; the bytes pem_tiled_save writes equal the bytes the
restatement writes from the oracle's arrays; C from cached tilings is bit-identical to C from converted ones.  Every
way a file can be wrong -- I/O, checksums, stale key, and checksum-valid but semantically invalid payloads -- is refused."""
import os

import numpy as np
import pytest

import cachefmt
from matgen import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["tiny_9x9", "one_entry", "rand_300", "powerlaw_600", "dense_tile", "rect_70x40_AAt", "ragged_37", "empty_rows", "wide_tilecols"]


def _oracle_file(oracle, tmp_path, name, transpose=False, key=(0, 0, 0)):
    rows, cols, I, J, V, _ = cases()[name]
    o = oracle.Tiled(rows, cols, I, J, V, transpose)
    path = str(tmp_path / f"{name}.pemtile")
    with open(path, "wb") as f:
        f.write(cachefmt.cache_bytes(o.rows, o.cols, o.tile_keys, o.tile_nnz_ptr, o.rowcolidx, o.vals, key))
    return path, o


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("transpose", [False, True])
def test_load_of_oracle_written_file_gives_oracle_arrays(pkg, oracle, ctx, tmp_path, name, transpose):
    path, o = _oracle_file(oracle, tmp_path, name, transpose)
    T = pkg.Tiled.load(ctx, path)
    assert (T.rows, T.cols, T.nnz, T.ntiles, T.tile_rows, T.tile_cols) == (o.rows, o.cols, o.nnz, o.ntiles, o.tile_rows, o.tile_cols)
    for arr in pkg.T_ARRAYS:
        assert np.array_equal(T.array(arr), getattr(o, arr)), arr


@pytest.mark.parametrize("name", NAMES)
def test_save_writes_the_restated_bytes(pkg, oracle, ctx, tmp_path, name):
    rows, cols, I, J, V, _ = cases()[name]
    key = pkg.CacheKey(123456, 987654321, 0, 0)
    T = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    path = str(tmp_path / "saved.pemtile")
    T.save(path, key)
    o = oracle.Tiled(rows, cols, I, J, V)
    want = cachefmt.cache_bytes(o.rows, o.cols, o.tile_keys, o.tile_nnz_ptr, o.rowcolidx, o.vals, (123456, 987654321, 0))
    assert open(path, "rb").read() == want
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]        # written via rename, nothing left behind


@pytest.mark.parametrize("name", ["tiny_9x9", "rand_50"])
def test_golden_cache_file_is_what_save_writes(pkg, ctx, tmp_path, name):
    """tests/golden/*.pemtile are committed; the CPU suite reads them with the restatement."""
    rows, cols, I, J, V, _ = cases()[name]
    T = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    path = str(tmp_path / "g.pemtile")
    T.save(path, None)
    assert open(path, "rb").read() == open(os.path.join(GOLD, name + ".pemtile"), "rb").read()
    L = pkg.Tiled.load(ctx, os.path.join(GOLD, name + ".pemtile"))
    for arr in pkg.T_ARRAYS:
        assert np.array_equal(L.array(arr), T.array(arr)), arr


@pytest.mark.parametrize("name", ["powerlaw_600", "rect_70x40_AAt"])
def test_spgemm_from_cached_tilings_is_bit_identical(pkg, ctx, tmp_path, name):
    rows, cols, I, J, V, _ = cases()[name]
    aat = rows != cols
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    B = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True) if aat else A
    pa, pb = str(tmp_path / "a.pemtile"), str(tmp_path / "b.pemtile")
    A.save(pa)
    B.save(pb)
    A2, B2 = pkg.Tiled.load(ctx, pa), pkg.Tiled.load(ctx, pb)
    p1, p2 = pkg.CPlan(ctx, A, B), pkg.CPlan(ctx, A2, B2)
    p1.spgemm()
    p2.spgemm()
    for arr in pkg.C_ARRAYS:
        assert np.array_equal(p1.array(arr), p2.array(arr)), arr
    for x, y in zip(p1.export_csr(), p2.export_csr()):
        assert np.array_equal(x, y)
    assert pkg.flop_count(ctx, A, B) == pkg.flop_count(ctx, A2, B2)


def test_key_round_trip_and_stale(pkg, ctx, tmp_path):
    rows, cols, I, J, V, _ = cases()["rand_300"]
    src = tmp_path / "m.mtx"
    src.write_text("placeholder source\n")
    key = pkg.CacheKey.of_file(str(src))
    T = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    path = str(tmp_path / "k.pemtile")
    T.save(path, key)
    assert cachefmt.read_cache(path)["key"] == (key.source_size, key.source_mtime_ns, 0)
    pkg.Tiled.load(ctx, path, key)                                  # same source: accepted
    pkg.Tiled.load(ctx, path, None)                                 # no expectation: accepted
    for other in (pkg.CacheKey(key.source_size + 1, key.source_mtime_ns, 0, 0), pkg.CacheKey(key.source_size, key.source_mtime_ns + 1, 0, 0),
                  pkg.CacheKey(key.source_size, key.source_mtime_ns, 1, 0)):
        with pytest.raises(pkg.PemError) as e:
            pkg.Tiled.load(ctx, path, other)
        assert e.value.status == -9                                 # PEM_E_STALE


def test_io_and_checksum_failures_are_refused(pkg, oracle, ctx, tmp_path):
    path, _ = _oracle_file(oracle, tmp_path, "rand_300")
    good = open(path, "rb").read()

    def refused(data, what):
        p = str(tmp_path / "bad.pemtile")
        with open(p, "wb") as f:
            f.write(data)
        with pytest.raises(pkg.PemError) as e:
            pkg.Tiled.load(ctx, p)
        assert e.value.status == -8, what                           # PEM_E_IO
    with pytest.raises(pkg.PemError) as e:
        pkg.Tiled.load(ctx, str(tmp_path / "absent.pemtile"))
    assert e.value.status == -8
    refused(b"", "empty file")
    refused(good[:100], "short header")
    refused(b"%%MatrixMarket matrix coordinate real general\n" + b" " * 200, "not a cache file")
    refused(good[:-64], "truncated payload")
    refused(good + b"\0", "trailing bytes")
    flip = bytearray(good)
    flip[128 + 40] ^= 0x10
    refused(bytes(flip), "payload bit flip")
    flip = bytearray(good)
    flip[33] ^= 0x01                                                # nnz field
    refused(bytes(flip), "header bit flip")
    with pytest.raises(pkg.PemError) as e:                          # unwritable target
        pkg.Tiled.load(ctx, path).save(str(tmp_path / "no_such_dir" / "x.pemtile"))
    assert e.value.status == -8
    pkg.Tiled.load(ctx, path)                                       # the context is still usable afterwards


def test_checksum_valid_but_invalid_payloads_are_refused(pkg, oracle, ctx, tmp_path):
    """A faulty (or hostile) producer: all checksums right, contents not a tiled matrix.  The device-side check must
    refuse each before any kernel indexes through the file's offsets."""
    rows, cols, I, J, V, _ = cases()["rand_300"]
    o = oracle.Tiled(rows, cols, I, J, V)
    base = dict(tile_keys=o.tile_keys.copy(), tile_nnz_ptr=o.tile_nnz_ptr.copy(), rowcolidx=o.rowcolidx.copy(), vals=o.vals.copy())
    T, nnz = o.ntiles, o.nnz
    assert T >= 4

    def attempt(what, rows_=o.rows, cols_=o.cols, nnz_=None, ntiles_=None, **changed):
        a = {k: v.copy() for k, v in base.items()}
        a.update(changed)
        p = str(tmp_path / "forged.pemtile")
        with open(p, "wb") as f:
            f.write(cachefmt.cache_bytes(rows_, cols_, a["tile_keys"], a["tile_nnz_ptr"], a["rowcolidx"], a["vals"], nnz=nnz_, ntiles=ntiles_))
        with pytest.raises(pkg.PemError) as e:
            pkg.Tiled.load(ctx, p)
        assert e.value.status == -8, what

    k = base["tile_keys"].copy(); k[[1, 2]] = k[[2, 1]]
    attempt("tile list not sorted", tile_keys=k)
    k = base["tile_keys"].copy(); k[2] = k[1]
    attempt("duplicate tile", tile_keys=k)
    k = base["tile_keys"].copy(); k[-1] = (np.int64(o.tile_rows) << 32) | 0
    attempt("tile row out of range", tile_keys=k)
    k = base["tile_keys"].copy(); k[0] = (k[0] & ~np.int64(0xFFFFFFFF)) | np.int64(o.tile_cols)
    attempt("tile column out of range", tile_keys=np.sort(k))
    k = base["tile_keys"].copy(); k[0] = -1
    attempt("negative key", tile_keys=k)
    q = base["tile_nnz_ptr"].copy(); q[0] = 1
    attempt("offsets do not start at 0", tile_nnz_ptr=q)
    q = base["tile_nnz_ptr"].copy(); q[-1] = nnz - 1
    attempt("offsets do not end at nnz", tile_nnz_ptr=q)
    q = base["tile_nnz_ptr"].copy(); q[2] = q[1]
    attempt("empty tile listed", tile_nnz_ptr=q)
    q = base["tile_nnz_ptr"].copy(); q[2] = 2 ** 30
    attempt("offset far beyond nnz", tile_nnz_ptr=q)
    q = base["tile_nnz_ptr"].copy(); q[2] = -5
    attempt("negative offset", tile_nnz_ptr=q)
    big = int(np.argmax(np.diff(base["tile_nnz_ptr"])))
    e0 = int(base["tile_nnz_ptr"][big])
    assert base["tile_nnz_ptr"][big + 1] - e0 >= 2
    r = base["rowcolidx"].copy(); r[e0 + 1] = r[e0]
    attempt("duplicate entry inside a tile", rowcolidx=r)
    r = base["rowcolidx"].copy(); r[[e0, e0 + 1]] = r[[e0 + 1, e0]]
    attempt("entries not row-major inside a tile", rowcolidx=r)
    # entry beyond the matrix edge: shrink the matrix under the same payload
    attempt("entry outside rows", rows_=int(np.max(I)))            # the entry in row max(I) no longer fits
    attempt("entry outside cols", cols_=int(np.max(J)))
    attempt("header counts disagree with payload (nnz)", nnz_=nnz + 1)
    attempt("header counts disagree with payload (ntiles)", ntiles_=T - 1)
    pkg.Tiled.load(ctx, _oracle_file(oracle, tmp_path, "rand_300")[0])   # still healthy


def test_empty_matrix_round_trip(pkg, ctx, tmp_path):
    e = np.zeros(0, np.int32)
    T = pkg.Tiled.from_coo(ctx, 40, 70, e, e, np.zeros(0))
    path = str(tmp_path / "empty.pemtile")
    T.save(path)
    L = pkg.Tiled.load(ctx, path)
    assert (L.rows, L.cols, L.nnz, L.ntiles) == (40, 70, 0, 0)
    for arr in ("tile_rowptr", "tile_colptr"):
        assert np.array_equal(L.array(arr), T.array(arr))


def test_cli_cache_skips_parse_and_conversion(pkg, standins, tmp_path):
    """`pemspgemm ... --cache DIR`: first run rebuilds and writes, second loads, an edited .mtx is rebuilt, a damaged
    cache file is rebuilt; C is the same file every time."""
    import importlib
    import subprocess
    hostio = importlib.import_module("pem_spgemm_amd.hostio")
    rows, cols, I, J, V = standins.make("scircuit", scale=0.02)
    mtx, cdir = str(tmp_path / "m.mtx"), tmp_path / "cache"
    cdir.mkdir()
    standins.write_mtx(mtx, rows, cols, I, J, V)
    env = dict(os.environ, PEM_CSV=str(tmp_path / "r.csv"), PEM_REPEAT="1")

    def run(tag, *extra):
        out_c = str(tmp_path / f"C_{tag}.mtx")
        out = subprocess.run([hostio.CLI_PATH, mtx, "0", *extra, "--cache", str(cdir), "--out", out_c], env=env, capture_output=True, text=True,
                             timeout=120)
        assert out.returncode == 0, out.stderr + out.stdout
        return out.stdout, open(out_c, "rb").read()

    so, c0 = run("cold")
    assert "0 loaded, 1 rebuilt" in so and (cdir / "m.mtx.A.pemtile").exists()
    so, c1 = run("warm")
    assert "1 loaded, 0 rebuilt" in so and c1 == c0
    so, c2 = run("aat", "1")                                          # A*A^T: A from the cache, A^T rebuilt and cached
    assert "1 loaded, 1 rebuilt" in so and (cdir / "m.mtx.AT.pemtile").exists()
    so, c3 = run("aat2", "1")
    assert "2 loaded, 0 rebuilt" in so and c3 == c2
    blob = bytearray((cdir / "m.mtx.A.pemtile").read_bytes())         # damaged cache: rebuilt, not trusted
    blob[300] ^= 0xFF
    (cdir / "m.mtx.A.pemtile").write_bytes(bytes(blob))
    so, c4 = run("damaged")
    assert "0 loaded, 1 rebuilt" in so and c4 == c0
    V2 = V.copy()                                                     # edited source (same shape): stale, rebuilt
    V2[0] += 1.0
    standins.write_mtx(mtx, rows, cols, I, J, V2)
    os.utime(mtx, ns=(1, 1))
    so, c5 = run("edited")
    assert "0 loaded, 1 rebuilt" in so and c5 != c0
