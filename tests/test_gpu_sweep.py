"""GPU: seeded random sweep over shapes, densities and operand combinations the fixed cases do not cover -- distinct A and
B (m x k times k x n), A*A^T of rectangular A, one-tile-wide and one-tile-tall operands, dense tiles next to empty
ones -- each against the serial Gustavson oracle (structure bit-exact, values bit-exact on the ascending-k fma chain)
and, for the tile-level arrays, against the tiled restatement."""
import numpy as np
import pytest

from prune_ref import expected

pytestmark = pytest.mark.gpu


def _rand(rng, rows, cols, nnz, clustered):
    nnz = min(nnz, rows * cols)
    if clustered:       # a few dense 16x16 neighbourhoods plus scatter: full tiles, long tile rows, empty tiles
        keys = set()
        for _ in range(max(1, nnz // 200)):
            r0, c0 = int(rng.integers(0, rows)), int(rng.integers(0, cols))
            rr = np.clip(r0 + rng.integers(-8, 9, 200), 0, rows - 1)
            cc = np.clip(c0 + rng.integers(-8, 9, 200), 0, cols - 1)
            keys.update((rr.astype(np.int64) * cols + cc).tolist())
        keys = np.fromiter(keys, dtype=np.int64)[:nnz]
    else:
        keys = rng.choice(rows * cols, nnz, replace=False)
    v = rng.uniform(-1, 1, len(keys))
    v[v == 0] = 0.5
    p = rng.permutation(len(keys))      # file order = shuffled
    return (keys[p] // cols).astype(np.int32), (keys[p] % cols).astype(np.int32), v[p]


SWEEP = []
_rng = np.random.default_rng(20260401)
for n in range(36):
    m, k, nn = (int(_rng.integers(1, 420)) for _ in range(3))
    dens = float(_rng.choice([0.002, 0.01, 0.05, 0.3]))
    SWEEP.append((n, m, k, nn, dens, bool(n % 3 == 0), ["ab", "aat", "aa"][n % 3 if n % 3 != 2 else 2]))


@pytest.mark.parametrize("seed,m,k,n,dens,clustered,mode", SWEEP, ids=[f"s{t[0]}_{t[6]}_{t[1]}x{t[2]}x{t[3]}" for t in SWEEP])
def test_random_products_match_the_oracle(pkg, oracle, ctx, seed, m, k, n, dens, clustered, mode):
    rng = np.random.default_rng(1000 + seed)
    if mode == "aa":
        k = m                                                       # square A for A*A
    AI, AJ, AV = _rand(rng, m, k, max(1, int(m * k * dens)), clustered)
    gA, oA, sA = pkg.Tiled.from_coo(ctx, m, k, AI, AJ, AV), oracle.Tiled(m, k, AI, AJ, AV), oracle.Csr(m, k, AI, AJ, AV)
    if mode == "ab":
        BI, BJ, BV = _rand(rng, k, n, max(1, int(k * n * dens)), not clustered)
        gB, oB, sB = pkg.Tiled.from_coo(ctx, k, n, BI, BJ, BV), oracle.Tiled(k, n, BI, BJ, BV), oracle.Csr(k, n, BI, BJ, BV)
    elif mode == "aat":
        gB, oB, sB = pkg.Tiled.from_coo(ctx, m, k, AI, AJ, AV, True), oracle.Tiled(m, k, AI, AJ, AV, True), oracle.Csr(m, k, AI, AJ, AV, True)
    else:
        gB, oB, sB = gA, oA, sA
    plan = pkg.CPlan(ctx, gA, gB)
    plan.spgemm()
    rp, ci, v = plan.export_csr()
    rp1, ci1, v1 = oracle.csr_spgemm(sA, sB).arrays()
    assert np.array_equal(rp, rp1) and np.array_equal(ci, ci1) and np.array_equal(v, v1)
    want, counts = expected(oracle.Plan(oA, oB), oA, oB)
    info = plan.info()
    assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == counts
    for arr in ("c_tile_rowptr", "c_tile_colidx", "pairs_offset", "pairs_a", "pairs_b", "c_mask", "c_tile_nnz_ptr", "c_rowcolidx", "c_vals"):
        assert np.array_equal(plan.array(arr), want[arr]), arr
    assert pkg.flop_count(ctx, gA, gB) == oracle.flop_count(oA, oB)
    plan.spgemm()                                                   # warm repeat pass: identical
    assert np.array_equal(plan.export_csr()[2], v)
