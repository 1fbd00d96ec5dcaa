"""GPU: round-3 host-side fixes -- step-wise calls after warm passes, the ticketed device scan, the context arena, plan options."""
import numpy as np
import pytest

from matgen import cases
from prune_ref import expected

pytestmark = pytest.mark.gpu

CASES = cases()
C_NAMES = ["c_tile_rowptr", "c_tile_rowidx", "c_tile_colidx", "pairs_offset", "pairs_a", "pairs_b", "c_mask", "c_tile_nnz_ptr",
           "c_rowptr", "c_rowcolidx", "c_vals"]


def _pair(pkg, oracle, ctx, case):
    rows, cols, I, J, V, tr = case
    gA = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, False)
    gB = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True) if tr else gA
    oA = oracle.Tiled(rows, cols, I, J, V, False)
    oB = oracle.Tiled(rows, cols, I, J, V, True) if tr else oA
    return gA, gB, oA, oB


@pytest.mark.parametrize("name", ["powerlaw_600", "hub_row_4000", "band_1500"])
@pytest.mark.parametrize("graph", [False, True])
def test_stepwise_calls_after_warm_passes(pkg, oracle, ctx, name, graph):
    """pem_spgemm_step2 / step3 may follow a warm (or graph-replayed) pem_spgemm on the same plan: step 2 then runs again on
    step 1's result of that pass and must not add onto the group counters the pass has already scanned (round-2 advisor
    finding: c_tile_nnz_ptr / c_rowcolidx came out wrong and step 3 produced garbage, status PEM_OK)."""
    gA, gB, oA, oB = _pair(pkg, oracle, ctx, CASES[name])
    plan = pkg.CPlan(ctx, gA, gB)
    ctx.set_graph_replay(graph)
    try:
        for _ in range(3):
            plan.spgemm()
        plan.step2()
        plan.step3()
        plan.step2()          # and once more: every step 2 clears what it accumulates into
        plan.step3()
    finally:
        ctx.set_graph_replay(False)
    want_arrays, want_counts = expected(oracle.Plan(oA, oB), oA, oB)
    info = plan.info()
    assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == want_counts
    for arr in C_NAMES:
        assert np.array_equal(plan.array(arr), want_arrays[arr]), f"{name}: {arr} differs after step-wise calls on a warm plan"
    plan.spgemm()             # ... and the plan is still good for whole passes
    for arr in C_NAMES:
        assert np.array_equal(plan.array(arr), want_arrays[arr]), f"{name}: {arr} differs on the pass after"


@pytest.mark.parametrize("n", [0, 1, 63, 2048, 2049, 8192, 8193, 70001, 262144])
def test_device_scan_regimes_agree(pkg, ctx, n):
    """the one-block, the chained single-launch and the three-launch scan give numpy's prefix sums, in place and out of place"""
    rng = np.random.default_rng(n + 1)
    v = rng.integers(0, 1000, n).astype(np.int32)
    want = np.concatenate([[0], np.cumsum(v, dtype=np.int64)]).astype(np.int32)
    for regime in (0, 1, 2, 3):
        for in_place in (True, False):
            got, total = ctx.debug_scan(v, regime=regime, in_place=in_place)
            assert total == int(v.sum()), (regime, in_place)
            assert np.array_equal(got, want), f"regime {regime} in_place {in_place}: scan of {n} items differs"


@pytest.mark.parametrize("stall", [0, 1, 17, 33])
def test_chained_scan_survives_a_stalled_block_in_place(pkg, ctx, stall):
    """Round-2 finding: the chained scan's fallback re-summed an input that in-place callers had already overwritten.  There
    is no fallback any more -- blocks draw tickets and wait only for tickets that are running.  One block is made to stall
    ~0.2 ms before it publishes (thousands of polls for everyone behind it); the in-place result must equal the one-block
    scan's."""
    n = 34 * 2048 + 77
    rng = np.random.default_rng(stall)
    v = rng.integers(0, 30000, n).astype(np.int32)
    ref, tref = ctx.debug_scan(v, regime=1, in_place=False)
    got, total = ctx.debug_scan(v, regime=2, in_place=True, stall_ticket=stall)
    assert total == tref == int(v.astype(np.int64).sum())
    assert np.array_equal(got, ref)
    again, _ = ctx.debug_scan(v, regime=2, in_place=True)          # the tickets were handed back: the next scan starts at 0
    assert np.array_equal(again, ref)


def test_scan_overflow_is_flagged_not_wrapped(pkg, ctx):
    """a total beyond int32 must reach the caller as 64 bits (the hot path turns it into PEM_E_OVERFLOW)"""
    v = np.full(3000, 2_000_000, dtype=np.int32)                    # 6e9
    for regime in (1, 2, 3):
        _, total = ctx.debug_scan(v, regime=regime, in_place=True)
        assert total == 6_000_000_000, regime


def test_arena_serves_a_second_plan_without_the_driver(pkg, oracle, ctx):
    """The context's arena keeps what a destroyed plan gives back: a fresh plan of the same product makes no driver
    allocation at all, and a first plan makes at most one per sizing phase (pairs / C tiles / C entries) beyond what the
    arena already holds."""
    rows, cols, I, J, V, tr = CASES["powerlaw_600"]
    gA = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    warm = pkg.CPlan(ctx, gA, gA)              # (the context's own temporaries -- scan state, export scratch -- grow on first use)
    warm.spgemm()
    warm.export_csr()
    warm.close()
    before = ctx.memory_stats()
    plan = pkg.CPlan(ctx, gA, gA)
    plan.spgemm()
    first = plan.export_csr()
    mid = ctx.memory_stats()
    assert mid["driver_allocs"] == before["driver_allocs"], "the arena already held what the same product gave back"
    assert mid["in_use_bytes"] > before["in_use_bytes"]
    plan.close()
    after_close = ctx.memory_stats()
    assert after_close["in_use_bytes"] == before["in_use_bytes"], "a destroyed plan must return every block to the arena"
    plan2 = pkg.CPlan(ctx, gA, gA)
    plan2.spgemm()
    second = plan2.export_csr()
    end = ctx.memory_stats()
    assert end["driver_allocs"] == mid["driver_allocs"], "the second plan must be served from the arena"
    for a, b in zip(first, second):
        assert np.array_equal(a, b)
    plan2.close()
    ctx.reserve(8 << 20)                       # a free block that large exists by now: no driver call
    assert ctx.memory_stats()["driver_allocs"] == end["driver_allocs"]
    ctx.trim()
    assert ctx.memory_stats()["slab_bytes"] <= end["slab_bytes"]
    plan3 = pkg.CPlan(ctx, gA, gA)             # trimming must leave the context usable
    plan3.spgemm()
    for a, b in zip(first, plan3.export_csr()):
        assert np.array_equal(a, b)


def test_plan_options_through_the_abi(pkg, oracle, ctx):
    """kernel variants are plan options (pem_cplan_set_option), not environment reads at call time: flipping them on a live
    plan restarts it and every variant still gives the oracle's arrays"""
    gA, gB, oA, oB = _pair(pkg, oracle, ctx, CASES["hub_row_4000"])
    op = oracle.Plan(oA, oB)
    pruned, pruned_counts = expected(op, oA, oB)
    unpruned, unpruned_counts = expected(op, oA, oB, False)
    plan = pkg.CPlan(ctx, gA, gB)
    assert plan.get_option("prune") == 1 and plan.get_option("wide") == 1 and plan.get_option("warm") == 1
    plan.spgemm()
    plan.spgemm()
    for opts, want, counts in (({"wide": 0}, pruned, pruned_counts), ({"wide": 1, "step1_global_sort": 1}, pruned, pruned_counts),
                               ({"step1_global_sort": 0, "prune": 0}, unpruned, unpruned_counts),
                               ({"prune": 1, "s1_xlcap": 40}, pruned, pruned_counts), ({"s1_xlcap": 0, "s1_force_key64": 1}, pruned, pruned_counts),
                               ({"s1_force_key64": 0, "warm": 0}, pruned, pruned_counts), ({"warm": 1, "s1_segments": 1}, pruned, pruned_counts),
                               ({"s1_segments": 0}, pruned, pruned_counts), ({"s2_transposed": 1}, pruned, pruned_counts),
                               ({"prune": 0}, unpruned, unpruned_counts), ({"prune": 1, "s2_transposed": 0}, pruned, pruned_counts),
                               ({"s2_transposed": 2}, pruned, pruned_counts)):
        for k, v in opts.items():
            plan.set_option(k, v)
            assert plan.get_option(k) == v
        for _ in range(2):
            plan.spgemm()
            info = plan.info()
            assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == counts, opts
            for arr in C_NAMES:
                assert np.array_equal(plan.array(arr), want[arr]), f"{opts}: {arr} differs"
    with pytest.raises(pkg.PemError):
        pkg._check(pkg.lib().pem_cplan_set_option(plan._h, 99, pkg.C.c_int64(1)))


@pytest.mark.parametrize("name", ["powerlaw_600", "hub_row_4000", "dense_48", "dense_tile", "ragged_37", "empty_rows", "banded_500_AAt", "rect_70x40_AAt",
                                  "rect_33x65_AAt", "cancel_and_zero", "band_1500"])
def test_step2_tile_product_from_transposed_masks(pkg, oracle, ctx, name):
    """PEM_OPT_S2_TRANSPOSED (include/pem_test.h): the boolean tile product C[r] |= B[k] taken column by column of the A tile -- over
    the inner indices occupied on both sides -- instead of nonzero by nonzero of its rows (spgemm.cu:499-550).  Every C array must
    equal the oracle's either way, pruned lists and the reference's own (where pairs with no common inner index give empty masks)."""
    gA, gB, oA, oB = _pair(pkg, oracle, ctx, CASES[name])
    op = oracle.Plan(oA, oB)
    plan = pkg.CPlan(ctx, gA, gB)
    for prune in (1, 0):
        want, counts = expected(op, oA, oB, bool(prune))
        plan.set_option("prune", prune)
        for tr in (1, 0):
            plan.set_option("s2_transposed", tr)
            for _ in range(2):
                plan.spgemm()
                for arr in C_NAMES:
                    assert np.array_equal(plan.array(arr), want[arr]), f"prune {prune} transposed {tr}: {arr} differs"


@pytest.mark.parametrize("name", ["powerlaw_600", "hub_row_4000", "dense_48", "dense_tile", "ragged_37", "empty_rows", "blockrows_1600", "rect_70x40_AAt"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_step3_reads_rows_and_columns_off_the_masks(pkg, oracle, ctx, name, dtype):
    """Shallow plans (fewer than two pairs per C tile) skip Ctiles_rowColIdx on the pass: step 3 finds an entry's row by a
    search over the tile's row prefixes and its column as the k-th set bit of the row mask (PEM_OPT_S3_DECODE, default on).
    C must be bit-identical to the variant that reads the bytes step 2 wrote, Ctiles_rowColIdx fetched afterwards (on demand)
    must be the oracle's, and a full 16x16 tile (dense_48: prefixes up to 240, entry 255) must decode."""
    rows, cols, I, J, V, tr = CASES[name]
    gA = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, False, dtype=dtype)
    gB = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True, dtype=dtype) if tr else gA
    plan = pkg.CPlan(ctx, gA, gB)
    assert plan.get_option("s3_decode") == 1
    plan.spgemm()
    plan.spgemm()
    v_dec = plan.array("c_vals")
    rc_dec = plan.array("c_rowcolidx")          # materialised now, from the masks
    csr_dec = plan.export_csr()
    plan.set_option("s3_decode", 0)
    plan.spgemm()
    v_ref = plan.array("c_vals")
    rc_ref = plan.array("c_rowcolidx")
    assert np.array_equal(rc_dec, rc_ref)
    assert np.array_equal(v_dec, v_ref), "values differ between the mask-decoding step 3 and the byte-reading one"
    for a, b in zip(csr_dec, plan.export_csr()):
        assert np.array_equal(a, b)
    # the decoding kernel's addressing (32-bit offsets / 64-bit indices) and entry -> tile lookup (marks / shuffle search)
    plan.set_option("s3_decode", 1)
    for idx64, mark in ((1, 1), (0, 0), (1, 0)):
        plan.set_option("s3_idx64", idx64)
        plan.set_option("s3_mark", mark)
        plan.spgemm()
        plan.spgemm()
        assert np.array_equal(plan.array("c_vals"), v_ref), f"idx64={idx64} mark={mark}: values differ"
    plan.set_option("s3_idx64", 0)
    plan.set_option("s3_mark", 1)
    if dtype == np.float64:
        oA = oracle.Tiled(rows, cols, I, J, V, False)
        oB = oracle.Tiled(rows, cols, I, J, V, True) if tr else oA
        want, _ = expected(oracle.Plan(oA, oB), oA, oB)
        assert np.array_equal(rc_dec, want["c_rowcolidx"]) and np.array_equal(v_dec, want["c_vals"])
    info = plan.info()
    deep = info["npairs"] >= 2 * info["ntiles_c"]
    ctx.set_kernel_profiling(True)
    ctx.reset_kernel_stats()
    plan.set_option("s3_decode", 1)
    plan.spgemm()
    names = list(ctx.kernel_stats())
    ctx.set_kernel_profiling(False)
    if not deep and info["ntiles_c"] > 0:      # the decode kernel really ran, and the entry kernel did not
        assert any("decode" in k for k in names) and "s2_entries_kernel" not in names, names


@pytest.mark.parametrize("name", ["blockrows_10000", "xl_mixed_AAt", "k64_bins", "hub_tile_2100_AAt", "blockrows_1600"])
@pytest.mark.parametrize("prune", [1, 0])
def test_big_rows_in_segments_and_one_workgroup_per_row(pkg, oracle, ctx, name, prune):
    """Tile rows above the 8192-key bin are cut into column-range segments, one workgroup each (s1_rowseg_kernel, round 4);
    PEM_OPT_S1_SEGMENTS = 0 keeps one workgroup per row (the 32768-key bin).  Both must give the oracle's lists, with
    32- and 64-bit keys, cold and warm -- hub_tile_2100_AAt drives the segments' range halving and their single-column emit."""
    gA, gB, oA, oB = _pair(pkg, oracle, ctx, CASES[name])
    want, counts = expected(oracle.Plan(oA, oB), oA, oB, bool(prune))
    for seg in (1, 0):
        for key64 in (0, 1):
            plan = pkg.CPlan(ctx, gA, gB)
            plan.set_option("prune", prune)
            plan.set_option("s1_segments", seg)
            plan.set_option("s1_force_key64", key64)
            ctx.set_kernel_profiling(True)
            ctx.reset_kernel_stats()
            for _ in range(2):
                plan.spgemm()
                info = plan.info()
                assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == counts, (seg, key64)
                for arr in C_NAMES:
                    assert np.array_equal(plan.array(arr), want[arr]), f"{name} segments {seg} key64 {key64} prune {prune}: {arr} differs"
            names = list(ctx.kernel_stats())
            ctx.set_kernel_profiling(False)
            if name in ("blockrows_10000", "xl_mixed_AAt", "hub_tile_2100_AAt"):          # rows above 8192 live products
                assert any(k.startswith("s1_rowseg_kernel") for k in names) == (seg == 1), names
            plan.close()


@pytest.mark.parametrize("name", ["hub_row_4000", "blockrows_10000", "xl_mixed_AAt", "k64_bins", "powerlaw_600"])
@pytest.mark.parametrize("prune", [1, 0])
def test_oversized_rows_sorted_per_row_and_globally(pkg, oracle, ctx, name, prune):
    """Tile rows beyond the LDS bins (here: forced by PEM_OPT_S1_XLCAP; on the webbase-1M stand-in its directory pages, whose
    4 700 A tiles fit no LDS table) are sorted one workgroup per row in global memory (s1_xl_rowsort_kernel) by default, and by
    the global (row, tile column) radix sort with PEM_OPT_S1_XL_GLOBAL.  Both must give the oracle's lists, cold and warm."""
    gA, gB, oA, oB = _pair(pkg, oracle, ctx, CASES[name])
    want, counts = expected(oracle.Plan(oA, oB), oA, oB, bool(prune))
    for cap in (40, 300):
        for glob in (0, 1):
            plan = pkg.CPlan(ctx, gA, gB)
            plan.set_option("prune", prune)
            plan.set_option("s1_xlcap", cap)
            plan.set_option("s1_xl_global", glob)
            ctx.set_kernel_profiling(True)
            ctx.reset_kernel_stats()
            for _ in range(2):
                plan.spgemm()
                info = plan.info()
                assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == counts, (cap, glob)
                for arr in C_NAMES:
                    assert np.array_equal(plan.array(arr), want[arr]), f"{name} cap {cap} global {glob} prune {prune}: {arr} differs"
            names = list(ctx.kernel_stats())
            ctx.set_kernel_profiling(False)
            if info["npairs"] > cap and any(k.startswith("s1_xl_gather") for k in names):
                assert ("s1_xl_rowsort_kernel" in names) == (glob == 0), names
                assert ("s1_xl_emit_kernel" in names) == (glob == 1), names
            plan.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_and_sparse_tiles_of_one_product(pkg, oracle, standins, ctx, dtype):
    """The round-3 webbase-1M stand-in (here 3 % of it) mixes nearly full C tiles (hosts whose pages all link to the index
    pages and back: prefixes up to 240, entry 255 in the mask decoding) with tiles of two or three entries.  CSR against the
    serial Gustavson oracle bit for bit (fp64), identical with the decoding switched off, and with 256 / 1024 entries per wave."""
    rows, cols, I, J, V = standins.make("webbase-1M", 0.03)
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, dtype=dtype)
    plan = pkg.CPlan(ctx, A, A)
    plan.spgemm()
    plan.spgemm()
    got = plan.export_csr()
    for opt, val in (("s3_decode", 0), ("s3_epw", 1), ("s3_epw", 4), ("s3_idx64", 1), ("s3_mark", 0), ("prune", 0), ("s3_xcd", 0)):
        for k, v in (("s3_decode", 1), ("s3_epw", 0), ("s3_idx64", 0), ("s3_mark", 1), ("prune", 1), ("s3_xcd", 1)):
            plan.set_option(k, v)
        plan.set_option(opt, val)
        plan.spgemm()
        for a, b in zip(got, plan.export_csr()):
            assert np.array_equal(a, b), (opt, val)
    if dtype == np.float64:
        oa = oracle.Csr(rows, cols, I, J, V)
        for a, b in zip(got, oracle.csr_spgemm(oa, oa).arrays()):
            assert np.array_equal(a, b)
    counts = np.bincount(np.repeat(np.arange(len(got[0]) - 1) // 16, np.diff(got[0])).astype(np.int64) * ((cols + 15) // 16) + got[1] // 16)
    assert (counts >= 64).sum() > 100 and ((counts > 0) & (counts < 8)).sum() > 100      # the input really has both kinds


def _special_values_matrix(n=700, seed=5):
    """Explicit zeros of both signs, subnormals, huge and tiny magnitudes, infinities and NaNs among ordinary values; a dense
    block so that several products meet in one entry, and sparse rows around it."""
    rng = np.random.default_rng(seed)
    dense = [(r, c) for r in range(40, 72) for c in range(40, 72) if rng.random() < 0.7]
    sparse = {(int(r), int(c)) for r, c in zip(rng.integers(0, n, 6000), rng.integers(0, n, 6000))}
    coords = sorted(set(dense) | sparse)
    I = np.array([p[0] for p in coords], dtype=np.int32)
    J = np.array([p[1] for p in coords], dtype=np.int32)
    V = rng.uniform(-1.0, 1.0, len(I))
    V[V == 0] = 0.5
    special = np.array([0.0, -0.0, 5e-324, -2.5e-310, 1e-200, -1e-200, 1e200, -1e200, 1.7e308, np.inf, -np.inf, np.nan])
    pick = rng.random(len(I)) < 0.08
    V[pick] = rng.choice(special, int(pick.sum()))
    return n, n, I, J, V


def _same_bits_or_both_nan(a, b):
    a, b = np.asarray(a), np.asarray(b)
    nan = np.isnan(a)
    u = np.uint64 if a.dtype == np.float64 else np.uint32
    return a.shape == b.shape and np.array_equal(nan, np.isnan(b)) and np.array_equal(a[~nan].view(u), b[~nan].view(u))


def test_special_values_follow_the_fma_chain(pkg, oracle, ctx):
    """Stored zeros (+0, -0), subnormals, 1e-200 .. 1.7e308, +-inf and NaN: C's structure is the structural product whatever
    the values (a stored zero still makes an entry), and every value is the oracle's chain -- the sign of a zero sum, the
    underflow to a subnormal, the overflow to inf included.  NaNs must sit where the oracle's sit (the payload of a NaN an
    instruction makes up is the hardware's)."""
    rows, cols, I, J, V = _special_values_matrix()
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    oa = oracle.Csr(rows, cols, I, J, V)
    want = oracle.csr_spgemm(oa, oa).arrays()
    assert np.isnan(want[2]).sum() > 10 and np.isinf(want[2]).sum() > 10 and (want[2] == 0).sum() > 10   # the input does what it says
    for opts in ({}, {"s3_decode": 0}, {"wide": 0}, {"prune": 0}, {"s3_epw": 1}):
        plan = pkg.CPlan(ctx, A, A)
        for k, v in opts.items():
            plan.set_option(k, v)
        plan.spgemm()
        plan.spgemm()                                   # (the repeat pass: graph-free warm plan)
        got = plan.export_csr()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), opts
        assert _same_bits_or_both_nan(got[2], want[2]), opts
        plan.close()
    op = oracle.Plan(oracle.Tiled(rows, cols, I, J, V, False), oracle.Tiled(rows, cols, I, J, V, False))
    assert _same_bits_or_both_nan(op.export_csr()[2], want[2])      # ... and the tiled oracle agrees with the serial one


def test_plans_come_and_go_under_graph_replay(pkg, standins, ctx):
    """Sixteen plans over changing row blocks of one product, each captured, replayed and closed before the next is made -- what
    multigpu.tune_row_bounds does.  The HIP runtime of this image crashed in hipGraphLaunch (hip::Graph::UpdateStreams) after
    8-12 such rounds while plans destroyed their graph executables; they are retired to the context instead.  Every block's C
    must also be the same whatever came before it."""
    rows, cols, I, J, V = standins.make("webbase-1M")
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    mt = A.tile_rows
    seen = {}
    ctx.set_graph_replay(True)
    try:
        for rnd in range(4):
            cuts = [0] + [int(mt * (k + 0.13 * ((rnd + k) % 3)) / 4) for k in range(1, 4)] + [mt]
            for p in range(4):
                plan = pkg.CPlan(ctx, A, A, cuts[p], cuts[p + 1])
                for _ in range(5):
                    plan.spgemm()
                info = plan.info()
                key = (cuts[p], cuts[p + 1])
                sig = (info["ntiles_c"], info["npairs"], info["nnz_c"], float(plan.array("c_vals").sum()))
                assert seen.setdefault(key, sig) == sig
                plan.close()
        whole = pkg.CPlan(ctx, A, A)
        whole.spgemm()
        whole.spgemm()
        assert whole.info()["nnz_c"] == 50187085
        whole.close()
    finally:
        ctx.set_graph_replay(False)
