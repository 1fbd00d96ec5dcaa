"""CPU: pins the oracle against the committed scipy fixtures (tests/golden/make_golden.py).
The reference has no golden vectors (PARITY UNPINNED at the reference boundary); these
fixtures come from an independent implementation."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "spgemm_*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-4] for p in FIXTURES])
def test_oracle_matches_scipy_fixture(oracle, path):
    z = _load(path)
    rows, cols, tr = int(z["rows"]), int(z["cols"]), bool(z["transpose"])
    I, J, V = z["I"], z["J"], z["V"]
    # independent serial Gustavson
    a, b = oracle.Csr(rows, cols, I, J, V, False), oracle.Csr(rows, cols, I, J, V, tr)
    rp, ci, v = oracle.csr_spgemm(a, b, 1).arrays()
    assert np.array_equal(rp, z["c_rowptr"]) and np.array_equal(ci, z["c_colidx"])
    # scipy sums in the same k order but without fma: agreement to a few ulp of the partial sums
    np.testing.assert_allclose(v, z["c_vals"], rtol=1e-12, atol=1e-14)
    # tiled restatement of the reference == serial CSR, bit for bit
    A = oracle.Tiled(rows, cols, I, J, V, False)
    B = oracle.Tiled(rows, cols, I, J, V, tr) if tr else A
    P = oracle.Plan(A, B)
    rp2, ci2, v2 = P.export_csr()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(v2, v)
    # OpenMP variant: same arithmetic, same bits
    rp3, ci3, v3 = oracle.csr_spgemm(a, b, 4).arrays()
    assert np.array_equal(rp3, rp) and np.array_equal(ci3, ci) and np.array_equal(v3, v)


def test_cage4_standin_is_the_baseline_config(oracle, standins):
    """BASELINE.json configs[0]: 9x9 / 49 nnz A^2 on the CPU serial CSR path, bit-exact vs the fixture structure."""
    z = _load(os.path.join(GOLD, "spgemm_cage4_standin.npz"))
    rows, cols, I, J, V = standins.cage4()
    assert (rows, cols, len(I)) == (9, 9, 49)
    assert np.array_equal(I, z["I"]) and np.array_equal(J, z["J"]) and np.array_equal(V, z["V"])
    a = oracle.Csr(rows, cols, I, J, V)
    rp, ci, v = oracle.csr_spgemm(a, a).arrays()
    assert np.array_equal(rp, z["c_rowptr"]) and np.array_equal(ci, z["c_colidx"])
    np.testing.assert_allclose(v, z["c_vals"], rtol=1e-13)


def test_tiled_layout_invariants(oracle):
    """layout facts of SURVEY 8(a) on a dense tile: u8 rowPtr max 240, slot 255, masks 0xFFFF"""
    z = _load(os.path.join(GOLD, "spgemm_dense_tile.npz"))
    T = oracle.Tiled(16, 16, z["I"], z["J"], z["V"])
    assert T.ntiles == 1 and T.tile_nnz_ptr.tolist() == [0, 256]
    assert T.masks.tolist() == [0xFFFF] * 16 and T.masks_t.tolist() == [0xFFFF] * 16
    assert T.rowptr.tolist() == [16 * r for r in range(16)]
    assert T.rowcolidx.tolist() == list(range(256))
    P = oracle.Plan(T, T)
    assert P.c_mask.tolist() == [0xFFFFFFFF] * 8 and P.nnz_c == 256 and P.c_rowptr.max() == 240


def test_duplicates_rejected(oracle):
    I = np.array([0, 0], np.int32)
    with pytest.raises(ValueError):
        oracle.Tiled(4, 4, I, I, np.array([1.0, 2.0]))
    with pytest.raises(ValueError):
        oracle.Csr(4, 4, I, I, np.array([1.0, 2.0]))


def test_row_slices_concatenate(oracle):
    """tile-row slices (the multi-GPU split) concatenate to the full result, structure and values"""
    z = _load(os.path.join(GOLD, "spgemm_powerlaw_600.npz"))
    rows, cols = int(z["rows"]), int(z["cols"])
    A = oracle.Tiled(rows, cols, z["I"], z["J"], z["V"])
    full = oracle.Plan(A, A).export_csr()
    parts = [oracle.Plan(A, A, lo, hi).export_csr() for lo, hi in ((0, 7), (7, 7), (7, 30), (30, A.tile_rows))]
    rp = np.concatenate([[0]] + [p[0][1:] + off for p, off in zip(parts, np.cumsum([0] + [len(p[1]) for p in parts[:-1]]))])
    assert np.array_equal(rp, full[0])
    assert np.array_equal(np.concatenate([p[1] for p in parts]), full[1])
    assert np.array_equal(np.concatenate([p[2] for p in parts]), full[2])
