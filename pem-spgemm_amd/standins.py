"""Seeded synthetic stand-ins for the SuiteSparse inputs BASELINE.json names.

No SuiteSparse file exists offline, so every config is generated.  `make(name)` goes through the C++
generator of libpemhost.so (host/standin.cpp, the one behind `pemspgemm --standin NAME`): round 3 calibrated the
webbase-1M and cage15 models so that the PRODUCT matches the literature (flop and C_nnz within 5 % of SURVEY 8(d)'s
figures), not only shape, nnz and degree skew.  The numpy generators below are the round-1/2 stand-ins, kept as
"<name>-r2" for continuity (they compress 1.02x / 1.08x where the real products compress 1.36x / 2.24x).  If a real
`.mtx` is present under --data, bench.py / the CLI use it instead.  Values are uniform in [-1, 1) excluding 0; no
duplicate (i, j); host-side only (data preparation, outside every timed region).
"""
import numpy as np

SIZES = {
    # name: (rows, cols, nnz)   [literature figures, SURVEY 8 -- unverified offline]
    "cage4": (9, 9, 49),
    "scircuit": (170998, 170998, 958936),
    "webbase-1M": (1000005, 1000005, 3105536),
    "mc2depi": (525825, 525825, 2100225),
    "cage15": (5154859, 5154859, 99199551),
}
SEEDS = {"cage4": 4, "scircuit": 171, "webbase-1M": 1000, "mc2depi": 526, "cage15": 5150}


def _values(rng, n):
    v = rng.uniform(-1.0, 1.0, n)
    v[v == 0.0] = 0.5
    return v


def _finish(rng, rows, cols, I, J, nnz):
    """dedupe (i, j), trim (or, rarely, top up) to exactly nnz entries, attach values, shuffle to 'file order'"""
    key = np.unique(I.astype(np.int64) * cols + J.astype(np.int64))
    while len(key) < nnz:   # top up with uniform entries (only when the structured draw fell short)
        extra = rng.integers(0, rows, 2 * (nnz - len(key))).astype(np.int64) * cols + rng.integers(0, cols, 2 * (nnz - len(key)))
        key = np.unique(np.concatenate([key, extra]))
    if len(key) > nnz:
        key = rng.choice(key, size=nnz, replace=False)
    key = rng.permutation(key)
    I = (key // cols).astype(np.int32)
    J = (key % cols).astype(np.int32)
    return rows, cols, I, J, _values(rng, nnz)


def _powerlaw_degrees(rng, n, total, alpha, dmax, dmin=1):
    """integer degrees ~ d^-alpha on [dmin, dmax], rescaled so that they sum to ~total"""
    u = rng.random(n)
    a = 1.0 - alpha
    d = (dmin ** a + u * (dmax ** a - dmin ** a)) ** (1.0 / a)
    d = d * (total / d.sum())
    d = np.clip(np.floor(d + rng.random(n)), 0, dmax).astype(np.int64)
    return d


def cage4(seed=SEEDS["cage4"]):
    """9x9, 49 nnz: full diagonal + 40 random off-diagonals (the real cage4 values are not available offline)."""
    rng = np.random.default_rng(seed)
    off = [(i, j) for i in range(9) for j in range(9) if i != j]
    pick = rng.choice(len(off), size=40, replace=False)
    I = np.array([i for i in range(9)] + [off[p][0] for p in pick], dtype=np.int32)
    J = np.array([i for i in range(9)] + [off[p][1] for p in pick], dtype=np.int32)
    order = rng.permutation(49)
    return 9, 9, I[order], J[order], _values(rng, 49)


def scircuit(seed=SEEDS["scircuit"], scale=1.0):
    """circuit-like: diagonal + Zipf(2.0) extra degree capped at 353; 60 % of columns within +-64 of the diagonal."""
    n0, _, nnz0 = SIZES["scircuit"]
    n, nnz = max(32, int(n0 * scale)), max(64, int(nnz0 * scale))
    rng = np.random.default_rng(seed)
    deg = _powerlaw_degrees(rng, n, int((nnz - n) * 1.03), 2.0, 353, 1)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    local = rng.random(len(rows)) < 0.6
    cols = np.where(local, rows + rng.integers(-64, 65, len(rows)), rng.integers(0, n, len(rows)))
    cols = np.clip(cols, 0, n - 1)
    I = np.concatenate([np.arange(n, dtype=np.int64), rows])
    J = np.concatenate([np.arange(n, dtype=np.int64), cols])
    return _finish(rng, n, n, I, J, nnz)


def webbase(seed=SEEDS["webbase-1M"], scale=1.0):
    """web-graph-like: power-law out-degree (alpha 2.1, max 4700); 90 % of a page's links stay in
    its neighbourhood (+-max(24, out-degree) rows: same host), 10 % go to globally popular pages
    whose popularity follows the out-degree (hubs link to hubs)."""
    n0, _, nnz0 = SIZES["webbase-1M"]
    n, nnz = max(64, int(n0 * scale)), max(128, int(nnz0 * scale))
    rng = np.random.default_rng(seed)
    deg = _powerlaw_degrees(rng, n, int(nnz * 1.25), 2.1, min(4700, n // 2), 1)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    m = len(rows)
    local = rng.random(m) < 0.9
    w = np.maximum(24, np.repeat(deg, deg))
    glob = rows[rng.integers(0, m, m)]   # endpoint of a random existing link: degree-proportional
    cols = np.where(local, rows + np.floor((rng.random(m) * 2.0 - 1.0) * (w + 1)).astype(np.int64), glob)
    cols = np.clip(cols, 0, n - 1)
    return _finish(rng, n, n, rows, cols, nnz)


def mc2depi(seed=SEEDS["mc2depi"], scale=1.0):
    """banded epidemiology-like: columns {i-1, i, i+1, i+floor(sqrt(n))} clipped; trimmed to the nnz."""
    n0, _, nnz0 = SIZES["mc2depi"]
    n = max(64, int(n0 * scale))
    nnz = nnz0 if scale == 1.0 else min(4 * n - 8, max(128, int(nnz0 * scale)))
    rng = np.random.default_rng(seed)
    w = int(np.sqrt(n))
    i = np.arange(n, dtype=np.int64)
    I = np.concatenate([i, i[1:], i[:-1], i[: n - w]])
    J = np.concatenate([i, i[1:] - 1, i[:-1] + 1, i[: n - w] + w])
    return _finish(rng, n, n, I, J, nnz)


def cage15(seed=SEEDS["cage15"], scale=1.0):
    """DNA-electrophoresis-like: ~19.2 nnz/row, columns i+delta with a two-scale mixture
    (90 % local +-500, 10 % uniform)."""
    n0, _, nnz0 = SIZES["cage15"]
    n, nnz = max(1024, int(n0 * scale)), max(4096, int(nnz0 * scale))
    rng = np.random.default_rng(seed)
    m = int(nnz * 1.04)
    rows = rng.integers(0, n, m).astype(np.int64)
    local = rng.random(m) < 0.9
    cols = np.where(local, rows + rng.integers(-500, 501, m), rng.integers(0, n, m))
    cols = np.clip(cols, 0, n - 1)
    return _finish(rng, n, n, rows, cols, nnz)


GENERATORS = {"cage4": cage4, "scircuit": scircuit, "webbase-1M": webbase, "mc2depi": mc2depi, "cage15": cage15}


def make(name, scale=1.0):
    """-> rows, cols, I, J, V (int32/int32/float64).  NAME: the C++ generator (sorted by row, column); NAME-r2: the
    round-2 numpy generator (file order = shuffled)"""
    if name.endswith("-r2"):
        base = name[:-3]
        if base == "cage4":
            return cage4()
        return GENERATORS[base](scale=scale)
    from . import hostio
    return hostio.standin(name, scale)


def write_mtx(path, rows, cols, I, J, V, comment="pem-spgemm_amd synthetic stand-in"):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"% {comment}\n")
        f.write(f"{rows} {cols} {len(I)}\n")
        for i, j, v in zip(I.tolist(), J.tolist(), V.tolist()):
            f.write(f"{i + 1} {j + 1} {v!r}\n")
