#!/usr/bin/env python3
"""Generates the committed golden fixtures with scipy (run in the build container only).

The reference ships no golden vectors and cannot run here, so these fixtures come
from an independent implementation: scipy.sparse products and scipy.io.mmread/mmwrite.
Outputs (all small):
  spgemm_<name>.npz   inputs (rows, cols, I, J, V, transpose) + expected C = A*B as CSR with
                      sorted columns: STRUCTURE from the pattern product (so numerically
                      cancelling entries stay), VALUES from scipy's float product.
  mm_<name>.mtx       Matrix-Market files exercising every field/symmetry the reader handles
  mm_<name>.npz       the COO scipy.io.mmread expands them to (sorted by (row, col))
"""
import os
import sys

import numpy as np
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from matgen import cases  # noqa: E402


def expected_product(rows, cols, I, J, V, tr):
    A = sp.coo_matrix((V, (I, J)), shape=(rows, cols)).tocsr()
    B = A.T.tocsr() if tr else A
    Ap, Bp = A.copy(), B.copy()
    Ap.data[:] = 1.0
    Bp.data[:] = 1.0
    S = (Ap @ Bp).tocsr()
    S.sort_indices()                       # structural product
    C = (A @ B).tocsr()
    C.sort_indices()
    dense_lookup = C.todok() if C.shape[0] * C.shape[1] < 1 << 22 else None
    vals = np.zeros(S.nnz)
    rws = np.repeat(np.arange(S.shape[0]), np.diff(S.indptr))
    Cc = C.tocoo()
    d = {(int(r), int(c)): float(v) for r, c, v in zip(Cc.row, Cc.col, Cc.data)}
    for n, (r, c) in enumerate(zip(rws, S.indices)):
        vals[n] = d.get((int(r), int(c)), 0.0)
    return S.indptr.astype(np.int32), S.indices.astype(np.int32), vals


def main():
    rng = np.random.default_rng(2025)
    for name, (rows, cols, I, J, V, tr) in cases().items():
        if name in ("wide_tilecols",):      # 262160^2 result index space: keep fixtures small
            continue
        rp, ci, v = expected_product(rows, cols, I, J, V, tr)
        np.savez_compressed(os.path.join(HERE, f"spgemm_{name}.npz"), rows=rows, cols=cols, I=I, J=J, V=V, transpose=int(tr),
                            c_rowptr=rp, c_colidx=ci, c_vals=v)
    # the 9x9 / 49-nnz cage4 stand-in of BASELINE.json configs[0]
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "pem-spgemm_amd"))
    import standins
    rows, cols, I, J, V = standins.cage4()
    rp, ci, v = expected_product(rows, cols, I, J, V, False)
    np.savez_compressed(os.path.join(HERE, "spgemm_cage4_standin.npz"), rows=rows, cols=cols, I=I, J=J, V=V, transpose=0, c_rowptr=rp,
                        c_colidx=ci, c_vals=v)
    scipy.io.mmwrite(os.path.join(HERE, "mm_cage4_standin.mtx"), sp.coo_matrix((V, (I, J)), shape=(rows, cols)), precision=17)

    # Matrix-Market reader fixtures
    def dump(name, text):
        path = os.path.join(HERE, f"mm_{name}.mtx")
        with open(path, "w") as f:
            f.write(text)
        M = scipy.io.mmread(path).tocoo()
        data = M.data.real if np.iscomplexobj(M.data) else M.data
        order = np.lexsort((M.col, M.row))
        np.savez_compressed(os.path.join(HERE, f"mm_{name}.npz"), rows=M.shape[0], cols=M.shape[1], I=M.row[order].astype(np.int32),
                            J=M.col[order].astype(np.int32), V=data[order].astype(np.float64))

    dump("general_real", "%%MatrixMarket matrix coordinate real general\n% comment\n\n4 5 6\n1 1 1.5\n2 3 -2.25e0\n4 5 1e-3\n3 1 7\n1 5 0.125\n4 2 -9.5E+1\n")
    dump("integer", "%%MatrixMarket matrix coordinate integer general\n3 3 4\n1 2 5\n2 1 -3\n3 3 12\n1 1 7\n")
    dump("pattern", "%%MatrixMarket matrix coordinate pattern general\n3 4 5\n1 1\n1 4\n2 2\n3 1\n3 3\n")
    dump("symmetric", "%%MatrixMarket matrix coordinate real symmetric\n4 4 5\n1 1 2.0\n2 1 -1.0\n3 2 0.5\n4 4 3.0\n4 1 8.0\n")
    dump("skew", "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 4.0\n3 1 -2.5\n")
    dump("complex", "%%MatrixMarket matrix coordinate complex general\n2 2 3\n1 1 1.0 2.0\n1 2 -3.5 0.0\n2 2 0.25 -1.0\n")
    dump("pattern_symmetric", "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n1 1\n3 1\n3 2\n")
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
