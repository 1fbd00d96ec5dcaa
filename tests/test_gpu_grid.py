"""GPU: 2-D (row block of A) x (column block of B) partition, emulated on one card: every block of the
grid is computed by its own pair of partial tilings -- no tiling ever holds all of A or of B -- and the spliced blocks
must equal the ORACLE's C (the serial CSR Gustavson port, float chain for fp32) bit for bit -- structure and values: each
entry still sums its full k range in order -- and, as a second check, the one-plan HIP result."""
import importlib

import numpy as np
import pytest
import torch

from matgen import cases

pytestmark = pytest.mark.gpu
CASES = cases()


@pytest.mark.parametrize("name,nrb,ncb", [("powerlaw_600", 2, 2), ("rect_70x40_AAt", 1, 3), ("blockrows_1600", 3, 2), ("rand_300", 4, 1),
                                          ("empty_rows", 2, 3), ("hub_row_4000", 2, 2)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_grid_blocks_splice_to_the_full_result(pkg, oracle, ctx, name, nrb, ncb, dtype):
    mg = importlib.import_module("pem_spgemm_amd.multigpu")
    rows, cols, I, J, V, tr = CASES[name]
    V = V.astype(dtype)
    BI, BJ = (J, I) if tr else (I, J)
    brows, bcols = (cols, rows) if tr else (rows, cols)
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, False, dtype=dtype)
    B = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True, dtype=dtype) if tr else A
    full = pkg.CPlan(ctx, A, B)
    full.spgemm()
    frp, fci, fv = full.export_csr()
    rb = mg.balanced_tile_bounds(I, rows, nrb)
    cb = mg.balanced_tile_bounds(BJ, bcols, ncb)
    assert rb[0] == 0 and rb[-1] == (rows + 15) // 16 and cb[-1] == (bcols + 15) // 16 and all(np.diff(rb) >= 0) and all(np.diff(cb) >= 0)
    blocks, held_a, held_b = [], 0, 0
    for i in range(nrb):
        ma = mg.restrict(I, rb[i], rb[i + 1])
        Ai = pkg.Tiled.from_coo(ctx, rows, cols, I[ma], J[ma], V[ma], False, dtype=dtype)
        held_a = max(held_a, Ai.nnz)
        for j in range(ncb):
            mb = mg.restrict(BJ, cb[j], cb[j + 1])
            Bj = pkg.Tiled.from_coo(ctx, brows, bcols, BI[mb], BJ[mb], V[mb], False, dtype=dtype)
            held_b = max(held_b, Bj.nnz)
            p = pkg.CPlan(ctx, Ai, Bj, rb[i], rb[i + 1])
            p.spgemm()
            blocks.append(tuple(torch.from_numpy(x) for x in p.export_csr()))
    rp, ci, v = mg.assemble_csr_blocks(blocks, ncb)
    # parity proper: against the CPU oracle on the same inputs (fp32: operands rounded to float, one fmaf per product)
    f32 = dtype == np.float32
    oa, ob = oracle.Csr(rows, cols, I, J, V.astype(np.float64), False), oracle.Csr(rows, cols, I, J, V.astype(np.float64), tr)
    orp, oci, ov = oracle.csr_spgemm(oa, ob, 1, f32=f32).arrays()
    assert np.array_equal(rp.numpy(), orp) and np.array_equal(ci.numpy(), oci)
    assert v.numpy().dtype == dtype and np.array_equal(v.numpy().astype(np.float64), ov)
    # ... and the one-plan result of the HIP path itself
    assert np.array_equal(rp.numpy(), frp) and np.array_equal(ci.numpy(), fci)
    assert np.array_equal(v.numpy(), fv)
    if nrb > 1 and len(I) > 1000:
        assert held_a < len(I)          # no block's tiling held all of A ...
    if ncb > 1 and len(I) > 1000:
        assert held_b < len(I)          # ... or all of B
