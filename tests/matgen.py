"""Small seeded matrices for the parity tests (edge cases of SURVEY 8(c))."""
import numpy as np


def _vals(rng, n):
    v = rng.uniform(-1.0, 1.0, n)
    v[v == 0.0] = 0.25
    return v


def random_coo(rng, rows, cols, nnz):
    key = rng.choice(rows * cols, size=min(nnz, rows * cols), replace=False)
    key = rng.permutation(key)
    return rows, cols, (key // cols).astype(np.int32), (key % cols).astype(np.int32), _vals(rng, len(key))


def cases():
    """name -> (rows, cols, I, J, V, transpose_B)"""
    rng = np.random.default_rng(12345)
    out = {}
    out["tiny_9x9"] = random_coo(rng, 9, 9, 49) + (False,)
    out["one_entry"] = (5, 5, np.array([3], np.int32), np.array([3], np.int32), np.array([2.5]), False)
    out["rand_50"] = random_coo(rng, 50, 50, 240) + (False,)
    out["rand_300"] = random_coo(rng, 300, 300, 1800) + (False,)
    out["rect_70x40_AAt"] = random_coo(rng, 70, 40, 230) + (True,)
    out["rect_33x65_AAt"] = random_coo(rng, 33, 65, 400) + (True,)
    # rows >= last multiple of 16, size not a multiple of 16
    out["ragged_37"] = random_coo(rng, 37, 37, 300) + (False,)
    # a fully dense 16x16 tile: 256 nnz, u8 rowPtr max 240, slot 255
    I, J = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    out["dense_tile"] = (16, 16, I.ravel().astype(np.int32), J.ravel().astype(np.int32), _vals(rng, 256), False)
    # dense 48x48 (9 dense tiles, every C tile dense, 3 pairs each)
    I, J = np.meshgrid(np.arange(48), np.arange(48), indexing="ij")
    p = rng.permutation(48 * 48)
    out["dense_48"] = (48, 48, I.ravel()[p].astype(np.int32), J.ravel()[p].astype(np.int32), _vals(rng, 48 * 48), False)
    # empty rows / empty tile rows / empty matrix
    r, c, I, J, V = random_coo(rng, 200, 200, 500)
    keep = (I < 40) | (I >= 120)
    out["empty_rows"] = (200, 200, I[keep], J[keep], V[keep], False)
    out["empty_matrix"] = (40, 40, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), False)
    # 1 x N and N x 1 shapes (A*A^T and A^T-like products)
    out["row_vector_AAt"] = (1, 100, np.zeros(30, np.int32), rng.choice(100, 30, replace=False).astype(np.int32), _vals(rng, 30), True)
    out["col_vector_AAt"] = (100, 1, rng.choice(100, 30, replace=False).astype(np.int32), np.zeros(30, np.int32), _vals(rng, 30), True)
    # explicit zero values and numerically cancelling products (entries must stay)
    I = np.array([0, 0, 1, 2, 1], np.int32)
    J = np.array([1, 2, 0, 0, 1], np.int32)
    V = np.array([1.0, 1.0, 3.0, -3.0, 0.0])
    out["cancel_and_zero"] = (3, 3, I, J, V, False)
    # power-law rows (webbase-like), banded (mc2depi-like), diagonal
    n = 600
    deg = np.minimum((rng.pareto(1.2, n) + 1).astype(int), 200)
    I = np.repeat(np.arange(n), deg)
    J = rng.integers(0, n, len(I))
    key = np.unique(I.astype(np.int64) * n + J)
    key = rng.permutation(key)
    out["powerlaw_600"] = (n, n, (key // n).astype(np.int32), (key % n).astype(np.int32), _vals(rng, len(key)), False)
    i = np.arange(500)
    I = np.concatenate([i, i[1:], i[:-1], i[:-22]])
    J = np.concatenate([i, i[1:] - 1, i[:-1] + 1, i[:-22] + 22])
    p = rng.permutation(len(I))
    out["banded_500_AAt"] = (500, 500, I[p].astype(np.int32), J[p].astype(np.int32), _vals(rng, len(I)), True)
    out["diag_100"] = (100, 100, np.arange(100, dtype=np.int32), np.arange(100, dtype=np.int32), _vals(rng, 100), False)
    # wide: more than 16384 tile columns (the reference's step-1 dispatch boundary, spgemm.cu:1142)
    out["wide_tilecols"] = random_coo(rng, 16385 * 16, 64, 900) + (True,)
    # tile rows with many products: 40 tiles x 40 tiles (1600 products/row: the four-wave LDS bin) and
    # 100 x 100 (10000 products/row: the 32768-key LDS bin; the global path with 64-bit keys), plus one hub row
    for nm, ntc in (("blockrows_1600", 40), ("blockrows_10000", 100)):
        n = 16 * ntc
        I = np.repeat(np.arange(n), ntc)
        J = (np.tile(np.arange(ntc), n) * 16 + (I * 7 + np.tile(np.arange(ntc), n) * 3) % 16)
        p = rng.permutation(len(I))
        out[nm] = (n, n, I[p].astype(np.int32), J[p].astype(np.int32), _vals(rng, len(I)), False)
    n = 4000
    I = np.concatenate([np.full(n, 5), np.arange(n), rng.integers(0, n, 6000)])
    J = np.concatenate([np.arange(n), np.arange(n), rng.integers(0, n, 6000)])
    key = rng.permutation(np.unique(I.astype(np.int64) * n + J))
    out["hub_row_4000"] = (n, n, (key // n).astype(np.int32), (key % n).astype(np.int32), _vals(rng, len(key)), False)
    # a band times itself (cage15-like): C tiles with 1..13 pairs and up to 256 entries -- step 3's many-pair kernel next to
    # the entry-per-lane one, pair counts that are no multiple of four, tiles of more than 64 entries
    n = 1500
    I = np.repeat(np.arange(n), 12)
    J = np.clip(I + rng.integers(-100, 101, len(I)), 0, n - 1)
    key = rng.permutation(np.unique(I.astype(np.int64) * n + J))
    out["band_1500"] = (n, n, (key // n).astype(np.int32), (key % n).astype(np.int32), _vals(rng, len(key)), False)
    out.update(step1_cases(rng))
    return out


def _grouped(rng, rows, ngroups, counts, hubs):
    """A (rows x 16*ngroups): `counts[k]` single entries in tile column k at distinct random rows, mostly in column
    16k+5 (so that products between tiles of one group are live) and a few elsewhere (so that pruning has work);
    hubs: list of (row, groups) -- rows holding an entry in each of the listed groups."""
    I, J = [], []
    for k, cnt in enumerate(counts):
        r = rng.choice(rows, size=cnt, replace=False)
        c = np.where(rng.random(cnt) < 0.85, 16 * k + 5, 16 * k + rng.integers(0, 16, cnt))
        I.append(r)
        J.append(c)
    for r, groups in hubs:
        I.append(np.full(len(groups), r))
        J.append(np.array([16 * k + 5 for k in groups]))
    I, J = np.concatenate(I).astype(np.int64), np.concatenate(J).astype(np.int64)
    key = rng.permutation(np.unique(I * (16 * ngroups) + J))
    return rows, 16 * ngroups, (key // (16 * ngroups)).astype(np.int32), (key % (16 * ngroups)).astype(np.int32), _vals(rng, len(key)), True


def step1_cases(rng):
    """The step-1 code a cage15-class input selects, reached WITHOUT test hooks."""
    out = {}
    # (i) B = A^T with 131 250 > 2^17 tile columns: keys need 64 bits.  A tile row of A*A^T with one entry in group k
    # has counts[k] products (bins <=512 and <=2048); hub rows with entries in many groups reach the <=8192 bin
    # (7 955 products) and go beyond it (9 855: above the largest LDS bin with 64-bit keys), so the global expand /
    # radix sort / emit path runs next to the LDS bins.
    counts = [215] * 37 + [100, 700, 1100]
    out["k64_bins"] = _grouped(rng, 2_100_000, 40, counts,
                               [(777_777, list(range(40))), (1_234_567, list(range(37))), (42, [0, 1, 2, 3, 38]), (2_099_999, [37, 39])])
    # (ii) 32-bit keys, tile rows above 32 768 live products next to rows of every LDS bin: A is 400 x 210 tiles,
    # tile rows 0..159 hold all 210 tiles (so every tile column of A = tile row of A^T has >= 160 tiles and a full
    # row has >= 33 600 products), the others hold 1..100 tiles.
    I, J = [], []
    for i in range(400):
        nt = 210 if i < 160 else (1, 2, 3, 10, 40, 100)[i % 6]
        ks = np.arange(210) if nt == 210 else rng.choice(210, size=nt, replace=False)
        I.append(np.full(nt, 16 * i + (i * 5) % 16))
        J.append(16 * ks + np.where(rng.random(nt) < 0.9, 5, rng.integers(0, 16, nt)))
    I, J = np.concatenate(I), np.concatenate(J)
    p = rng.permutation(len(I))
    out["xl_mixed_AAt"] = (6400, 3360, I[p].astype(np.int32), J[p].astype(np.int32), _vals(rng, len(I)), True)
    # (iii) C tiles of more than 2048 PAIRS and big rows whose keys crowd a few tile columns: five rows of A (in tile rows 0 .. 4
    # of 4096) hold an entry in each of 2100 tile columns, so in A*A^T every one of the 25 C tiles between them has 2100 pairs and
    # each of the five tile rows 10 500 live products in five tile columns out of 4096 -- the column-range segments of round 4 must
    # halve their ranges down to single columns and emit those unsorted
    hub_rows = np.array([3, 20, 40, 50, 70])
    I = np.repeat(hub_rows, 2100)
    J = np.tile(16 * np.arange(2100) + 5, 5)
    extra_i, extra_j = rng.integers(0, 65536, 500), rng.integers(0, 33600, 500)
    key = rng.permutation(np.unique(np.concatenate([I.astype(np.int64) * 33600 + J, extra_i.astype(np.int64) * 33600 + extra_j])))
    out["hub_tile_2100_AAt"] = (65536, 33600, (key // 33600).astype(np.int32), (key % 33600).astype(np.int32), _vals(rng, len(key)), True)
    return out
