"""CPU: the pemspgemm tool's host I/O (libpemhost.so) -- Matrix-Market reader against scipy.io
fixtures and the oracle reader; result-file and CSV formats against spgemm.cu:1424-1450, 1527-1560."""
import glob
import importlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MM = sorted(p for p in glob.glob(os.path.join(GOLD, "mm_*.mtx")) if os.path.exists(p[:-4] + ".npz"))


@pytest.fixture(scope="module")
def hostio(pkg):
    return importlib.import_module("pem_spgemm_amd.hostio")


def _sorted(m):
    order = np.lexsort((m["J"], m["I"]))
    return m["I"][order], m["J"][order], m["V"][order]


@pytest.mark.parametrize("path", MM, ids=[os.path.basename(p)[3:-4] for p in MM])
def test_mm_readers_match_scipy(hostio, oracle, path):
    z = np.load(path[:-4] + ".npz", allow_pickle=False)
    for reader in (hostio.mm_read, oracle.mm_read):
        m = reader(path)
        assert (m["rows"], m["cols"], m["nnz"]) == (int(z["rows"]), int(z["cols"]), len(z["I"]))
        I, J, V = _sorted(m)
        assert np.array_equal(I, z["I"]) and np.array_equal(J, z["J"]) and np.array_equal(V, z["V"])


def test_mm_roundtrip_and_threads(hostio, oracle, standins, tmp_path):
    rows, cols, I, J, V = standins.make("scircuit", scale=0.2)   # ~190k lines: exercises the chunked parallel parse
    p = str(tmp_path / "m.mtx")
    standins.write_mtx(p, rows, cols, I, J, V)
    for threads in (1, 5):
        m = hostio.mm_read(p, threads)
        assert (m["rows"], m["cols"]) == (rows, cols)
        assert np.array_equal(m["I"], I) and np.array_equal(m["J"], J) and np.array_equal(m["V"], V)   # file order, exact doubles
    mo = oracle.mm_read(p)
    assert np.array_equal(mo["I"], I) and np.array_equal(mo["V"], V)


def test_mm_errors(hostio, tmp_path):
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(RuntimeError):
        hostio.mm_read(str(bad))
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n")
    with pytest.raises(RuntimeError):
        hostio.mm_read(str(bad))
    with pytest.raises(RuntimeError):
        hostio.mm_read(str(tmp_path / "missing.mtx"))


def test_result_files_format(hostio, tmp_path):
    hostio.write_result_files(str(tmp_path), [0, 0, 7], [1, 5, 7], [1.5, -0.25, 1e-3])
    assert (tmp_path / "SPGEMM_RESULT_NNZ.txt").read_text() == "3"                       # no newline (spgemm.cu:1546)
    assert (tmp_path / "SPGEMM_RESULT_ROWS.txt").read_text() == "0\n0\n7\n"
    assert (tmp_path / "SPGEMM_RESULT_COLS.txt").read_text() == "1\n5\n7\n"
    assert (tmp_path / "SPGEMM_RESULT_VALS.txt").read_text().splitlines() == ["%.17f" % v for v in (1.5, -0.25, 1e-3)]


def test_csv_record_format(hostio, tmp_path):
    p = str(tmp_path / "r.csv")
    kw = dict(matrix="webbase-1M", flop=69524195, c_nnz=51111996, compression_ratio=1.3602, a_conversion_kernel_ms=1.234,
              b_conversion_kernel_ms=2.0, total_conversion_ms=300.456, step1_ms=1.0, step2_ms=2.5, step3_ms=3.25, spgemm_ms=7.0,
              kernel_ms=6.75, malloc_ms=0.25, gflops=19.864)
    hostio.csv_append(p, **kw)
    hostio.csv_append(p, extra="1,42", **kw)
    text = open(p).read()
    want = "\nwebbase-1M,69524195,51111996,1.36,1.23,2.00,300.46,1.00,2.50,3.25,7.00,6.75,0.25,19.86"
    assert text == want + want + ",1,42"   # "\n" + 14 fields, no header (spgemm.cu:1432-1448)


def test_mtx_csr_writer_roundtrip(hostio, oracle, tmp_path):
    import scipy.io
    rng = np.random.default_rng(7)
    rows, cols = 23, 31
    key = rng.choice(rows * cols, 120, replace=False)
    I, J, V = (key // cols).astype(np.int32), (key % cols).astype(np.int32), rng.uniform(-1, 1, 120)
    rp, ci, v = oracle.Csr(rows, cols, I, J, V).arrays()
    p = str(tmp_path / "c.mtx")
    hostio.write_mtx_csr(p, rows, cols, rp, ci, v, "roundtrip")
    M = scipy.io.mmread(p).tocsr()
    M.sort_indices()
    assert M.shape == (rows, cols) and np.array_equal(M.indptr, rp) and np.array_equal(M.indices, ci)
    assert np.array_equal(M.data, v)                      # 17 significant digits: exact doubles
    back = hostio.mm_read(p)
    assert back["nnz"] == 120 and np.array_equal(back["V"], v)


@pytest.mark.parametrize("seed", range(12))
def test_mm_reader_sweep_against_scipy(hostio, oracle, tmp_path, seed):
    """seeded sweep over the header variants and text quirks a Matrix-Market file may legally carry -- field real /
    integer / pattern, symmetry general / symmetric / skew-symmetric, comment and blank lines, tabs and trailing blanks,
    exponents, upper-case keywords -- with scipy.io.mmread (scipy 1.15's reader IS fast_matrix_market, the library the
    reference parses with: spgemm.cu:60-81) as the judge of what the file means"""
    import scipy.io
    rng = np.random.default_rng(7000 + seed)
    field = ["real", "integer", "pattern"][seed % 3]
    symm = ["general", "symmetric", "skew-symmetric"][(seed // 3) % 3]
    if field == "pattern" and symm == "skew-symmetric":
        symm = "symmetric"                                   # the format has no skew pattern matrices
    n = int(rng.integers(3, 60))
    m = n if symm != "general" else int(rng.integers(3, 60))
    keys = rng.choice(n * m, int(rng.integers(1, max(2, n * m // 3))), replace=False)
    I, J = keys // m, keys % m
    if symm != "general":
        keep = I > J if symm == "skew-symmetric" else I >= J        # stored triangle only
        I, J = I[keep], J[keep]
        if len(I) == 0:
            I, J = np.array([n - 1]), np.array([0])
    vals = rng.uniform(-50, 50, len(I))
    lines = ["%%MatrixMarket" + f" matrix coordinate {field.upper() if seed % 4 == 1 else field} {symm}",
             "% a comment", "%", f"{n} {m} {len(I)}" + ("  " if seed % 2 else "")]
    for r, (i, j, v) in enumerate(zip(I, J, vals)):
        sep = "\t" if (r + seed) % 5 == 0 else " " * (1 + (r % 3))
        if field == "pattern":
            lines.append(f"{i + 1}{sep}{j + 1}")
        elif field == "integer":
            lines.append(f"{i + 1}{sep}{j + 1}{sep}{int(v) if int(v) else 3}")
        else:
            lines.append(f"{i + 1}{sep}{j + 1}{sep}" + (f"{v:.17e}" if r % 2 else repr(float(v))) + (" " if r % 7 == 0 else ""))
        if r == 2:
            lines.append("")                                  # a blank line inside the data
    p = tmp_path / f"s{seed}.mtx"
    # (no final newline only for general files: scipy 1.15's reader crashes on a skew-symmetric file without one)
    p.write_text("\n".join(lines) + ("\n" if seed % 3 or symm != "general" else ""))
    want = scipy.io.mmread(str(p)).tocsr()
    want.sort_indices()
    for reader in (hostio.mm_read, oracle.mm_read):
        got = reader(str(p))
        assert (got["rows"], got["cols"]) == want.shape
        import scipy.sparse as sp
        G = sp.coo_matrix((got["V"], (got["I"], got["J"])), shape=want.shape)
        assert G.nnz == len(got["V"])
        G = G.tocsr()                                          # would sum duplicates: the nnz check below catches any
        G.sort_indices()
        assert G.nnz == want.nnz and np.array_equal(G.indptr, want.indptr) and np.array_equal(G.indices, want.indices)
        assert np.array_equal(G.data, want.data), (field, symm)


def test_bench_data_dir_uses_the_real_file_when_present(pkg, standins, tmp_path):
    """bench.py --data DIR: DIR/<workload>.mtx is read through pem_mm_read and reported as "real"; without the file the
    seeded stand-in is used (real SuiteSparse files are not in the container)."""
    import argparse
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows, cols, I, J, V = standins.make("scircuit", 0.002)
    standins.write_mtx(str(tmp_path / "scircuit.mtx"), rows, cols, I, J, V)
    args = argparse.Namespace(data=str(tmp_path), workload="scircuit", scale=0.002)
    r, c, i2, j2, v2, source, _ = bench.produce_input(args)
    assert source == "real" and (r, c, len(i2)) == (rows, cols, len(I))
    assert np.array_equal(i2, I) and np.array_equal(j2, J) and np.array_equal(v2, V)      # file order, exact doubles
    args = argparse.Namespace(data=str(tmp_path), workload="mc2depi", scale=0.002)        # no such file under DIR
    assert bench.produce_input(args)[5] == "synthetic"
