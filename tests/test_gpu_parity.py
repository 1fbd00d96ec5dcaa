"""GPU parity: the HIP path through the C ABI vs the CPU oracle, stage by stage, bit-exact."""
import numpy as np
import pytest

from matgen import cases
from prune_ref import expected

pytestmark = pytest.mark.gpu

CASES = cases()
T_NAMES = ["tile_keys", "tile_nnz_ptr", "masks", "rowptr", "rowcolidx", "vals", "masks_t", "tile_rowptr", "tile_colidx",
           "tile_colptr", "tile_rowidx", "tile_offsets"]
C_NAMES = ["c_tile_rowptr", "c_tile_rowidx", "c_tile_colidx", "pairs_offset", "pairs_a", "pairs_b", "c_mask", "c_tile_nnz_ptr",
           "c_rowptr", "c_rowcolidx", "c_vals"]


_ORACLE_TILED = {}     # the big step-1 cases take the oracle ~15 s to tile: once per session is enough


def _tiled_pair(pkg, oracle, ctx, case):
    rows, cols, I, J, V, tr = case
    gA = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, False)
    gB = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True) if tr else gA
    key = (rows, cols, len(I), int(I.sum()) if len(I) else 0, tr)
    if key not in _ORACLE_TILED:
        oA = oracle.Tiled(rows, cols, I, J, V, False)
        _ORACLE_TILED[key] = (oA, oracle.Tiled(rows, cols, I, J, V, True) if tr else oA)
    oA, oB = _ORACLE_TILED[key]
    return gA, gB, oA, oB


_ORACLE_PLAN = {}


def _oracle_plan(oracle, oA, oB):
    key = (id(oA), id(oB))
    if key not in _ORACLE_PLAN:
        _ORACLE_PLAN[key] = oracle.Plan(oA, oB)
    return _ORACLE_PLAN[key]


@pytest.mark.parametrize("name", list(CASES))
def test_tiled_conversion_matches_oracle(pkg, oracle, ctx, name):
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
    for g, o in ((gA, oA), (gB, oB)):
        assert (g.rows, g.cols, g.nnz, g.tile_rows, g.tile_cols, g.ntiles) == (o.rows, o.cols, o.nnz, o.tile_rows, o.tile_cols, o.ntiles)
        for arr in T_NAMES:
            got, want = g.array(arr), getattr(o, arr)
            assert got.dtype == want.dtype and np.array_equal(got, want), f"{name}: tiled array {arr} differs"


@pytest.mark.parametrize("name", list(CASES))
def test_three_steps_match_oracle(pkg, oracle, ctx, name):
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
    plan = pkg.CPlan(ctx, gA, gB)
    plan.step1()
    plan.step2()
    plan.step3()
    op = _oracle_plan(oracle, oA, oB)
    info = plan.info()
    want_arrays, want_counts = expected(op, oA, oB)      # the oracle's arrays minus dead pairs / empty tiles
    assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == want_counts
    for arr in C_NAMES:
        got, want = plan.array(arr), want_arrays[arr]
        assert got.dtype == want.dtype and np.array_equal(got, want), f"{name}: plan array {arr} differs"
    assert pkg.flop_count(ctx, gA, gB) == oracle.flop_count(oA, oB)
    # a14: CSR / sorted COO, against the tiled oracle and the independent serial CSR Gustavson
    rp, ci, v = plan.export_csr()
    rp0, ci0, v0 = op.export_csr()
    assert np.array_equal(rp, rp0) and np.array_equal(ci, ci0) and np.array_equal(v, v0)
    rows, cols, I, J, V, tr = CASES[name]
    sa, sb = oracle.Csr(rows, cols, I, J, V, False), oracle.Csr(rows, cols, I, J, V, tr)
    rp1, ci1, v1 = oracle.csr_spgemm(sa, sb).arrays()
    assert np.array_equal(rp, rp1) and np.array_equal(ci, ci1)
    assert np.array_equal(v, v1), "values must match the ascending-k fma chain bit for bit"
    r, c, vv = plan.export_coo()
    r0, c0, vv0 = op.export_coo()
    assert np.array_equal(r, r0) and np.array_equal(c, c0) and np.array_equal(vv, vv0)


def test_one_shot_spgemm_and_rerun(pkg, oracle, ctx):
    rows, cols, I, J, V, tr = CASES["powerlaw_600"]
    gA = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    plan = pkg.CPlan(ctx, gA, gA)
    plan.spgemm()
    first = plan.export_csr()
    plan.spgemm()   # plan re-use: buffers are recycled, results identical
    second = plan.export_csr()
    for a, b in zip(first, second):
        assert np.array_equal(a, b)
    t = ctx.timings()
    assert t["spgemm_wall_ms"] > 0 and t["step3_ms"] >= 0


@pytest.mark.parametrize("name", ["powerlaw_600", "blockrows_10000", "dense_48", "empty_matrix"])
def test_global_sort_step1_matches_row_local_step1(pkg, oracle, ctx, name, monkeypatch):
    """PEM_STEP1=esc selects the global expand/radix-sort step 1 (also the path of oversized rows)."""
    monkeypatch.setenv("PEM_STEP1", "esc")
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
    plan = pkg.CPlan(ctx, gA, gB)
    plan.spgemm()
    want_arrays, _ = expected(oracle.Plan(oA, oB), oA, oB)
    for arr in C_NAMES:
        assert np.array_equal(plan.array(arr), want_arrays[arr]), f"{name}: plan array {arr} differs (esc)"


@pytest.mark.parametrize("name", ["powerlaw_600", "dense_48", "rect_70x40_AAt", "hub_row_4000", "empty_matrix"])
def test_narrow_step23_kernels_match(pkg, oracle, ctx, name, monkeypatch):
    """PEM_WIDE=0 selects the 16-lanes-per-tile step 2/3 kernels (kept as the A/B baseline)."""
    monkeypatch.setenv("PEM_WIDE", "0")
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
    plan = pkg.CPlan(ctx, gA, gB)
    plan.spgemm()
    want_arrays, _ = expected(oracle.Plan(oA, oB), oA, oB)
    for arr in C_NAMES:
        assert np.array_equal(plan.array(arr), want_arrays[arr]), f"{name}: plan array {arr} differs (narrow)"


def test_repeat_passes_skip_readbacks_and_stay_identical(pkg, oracle, ctx, monkeypatch):
    """pem_spgemm on an unchanged plan re-uses the sizes of the previous pass (no host read-backs, device-side
    check); results must stay identical, also against a forced size-reading pass (PEM_OPT_WARM = 0)."""
    from matgen import cases as _cases
    for name in ("powerlaw_600", "hub_row_4000", "empty_matrix", "blockrows_10000"):
        gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
        plan = pkg.CPlan(ctx, gA, gB)
        plan.spgemm()                      # cold
        cold = [plan.array(a) for a in C_NAMES]
        for _ in range(3):
            plan.spgemm()                  # warm
        warm = [plan.array(a) for a in C_NAMES]
        plan.set_option("warm", 0)         # every pass reads its sizes back (the environment is only read at plan creation)
        plan.spgemm()
        forced = [plan.array(a) for a in C_NAMES]
        plan.set_option("warm", 1)
        want_arrays, _ = expected(oracle.Plan(oA, oB), oA, oB)
        for a, c, w, f in zip(C_NAMES, cold, warm, forced):
            want = want_arrays[a]
            assert np.array_equal(c, want) and np.array_equal(w, want) and np.array_equal(f, want), (name, a)


@pytest.mark.parametrize("name", ["powerlaw_600", "hub_row_4000", "ragged_37", "empty_rows"])
def test_row_serial_export_matches(pkg, oracle, ctx, name, monkeypatch):
    """PEM_OPT_EXPORT_ROWS selects the 16-lanes-per-tile-row export (A/B baseline of the chunked export)."""
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
    plan = pkg.CPlan(ctx, gA, gB)
    plan.spgemm()
    fast = plan.export_csr()
    plan.set_option("export_rows", 1)
    assert plan.get_option("export_rows") == 1
    plan.spgemm()                         # (changing an option restarts the plan: the next pass is a full one)
    slow = plan.export_csr()
    want = oracle.Plan(oA, oB).export_csr()
    for a, b, c in zip(fast, slow, want):
        assert np.array_equal(a, c) and np.array_equal(b, c)


@pytest.mark.parametrize("name", list(CASES))
def test_unpruned_lists_are_the_reference_lists(pkg, oracle, ctx, name, monkeypatch):
    """PEM_PRUNE=0: every tile-level product is kept, so C tile list, pair lists and every count equal the
    reference-faithful oracle's exactly -- for the row-local and for the global step 1."""
    monkeypatch.setenv("PEM_PRUNE", "0")
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
    op = _oracle_plan(oracle, oA, oB)
    for mode in ("rows", "esc"):
        monkeypatch.setenv("PEM_STEP1", mode)
        plan = pkg.CPlan(ctx, gA, gB)
        plan.spgemm()
        info = plan.info()
        assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == (op.ntiles_c, op.npairs, op.nnz_c, op.npairs)
        for arr in C_NAMES:
            assert np.array_equal(plan.array(arr), getattr(op, arr)), f"{name}/{mode}: plan array {arr} differs (unpruned)"


HOOKS = {"key64": {"PEM_S1_FORCE_KEY64": "1"}, "xl300": {"PEM_S1_XLCAP": "300"},
         "key64+xl300": {"PEM_S1_FORCE_KEY64": "1", "PEM_S1_XLCAP": "300"}, "xl40": {"PEM_S1_XLCAP": "40"}}


@pytest.mark.parametrize("prune", ["1", "0"])
@pytest.mark.parametrize("hook", list(HOOKS))
@pytest.mark.parametrize("name", ["powerlaw_600", "hub_row_4000", "blockrows_1600", "blockrows_10000", "dense_48", "rect_70x40_AAt",
                                  "wide_tilecols", "xl_mixed_AAt"])
def test_step1_wide_key_and_oversized_row_paths(pkg, oracle, ctx, name, hook, prune, monkeypatch):
    """The step-1 code a cage15-class input selects, forced onto small inputs: PEM_S1_FORCE_KEY64=1 runs the uint64-key
    row sorts (taken when B has more than 2^17 tile columns; with them the largest LDS bin is 8192 keys, so
    blockrows_10000 also takes the 64-bit global path), PEM_S1_XLCAP=n sends every tile row above n live products
    through s1_xl_expand -> radix sort -> s1_xl_rowstart -> s1_xl_emit next to rows that stay in the LDS bins.  All
    step-1/2/3 arrays against the oracle, pruned and unpruned (= the reference's own lists)."""
    for k, v in HOOKS[hook].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PEM_PRUNE", prune)
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES[name])
    op = _oracle_plan(oracle, oA, oB)
    want_arrays, want_counts = expected(op, oA, oB, prune == "1")
    plan = pkg.CPlan(ctx, gA, gB)
    for rnd in range(2):                      # cold pass, then a warm one (sizes re-used, no read-backs)
        plan.spgemm()
        info = plan.info()
        assert (info["ntiles_c"], info["npairs"], info["nnz_c"], info["npairs_all"]) == want_counts
        for arr in C_NAMES:
            assert np.array_equal(plan.array(arr), want_arrays[arr]), f"{name}/{hook}/prune={prune}/pass {rnd}: {arr} differs"


def test_options_are_latched_at_plan_creation(pkg, oracle, ctx, monkeypatch):
    """PEM_PRUNE changes the sizes of a pass; a warm plan re-uses sizes, so the option is read once, when the plan is
    made: flipping the environment under a live plan must not change (or corrupt) what it computes."""
    gA, gB, oA, oB = _tiled_pair(pkg, oracle, ctx, CASES["powerlaw_600"])
    op = _oracle_plan(oracle, oA, oB)
    plan = pkg.CPlan(ctx, gA, gB)
    plan.spgemm()
    plan.spgemm()
    monkeypatch.setenv("PEM_PRUNE", "0")
    plan.spgemm()                                         # still the pruned lists
    want_arrays, want_counts = expected(op, oA, oB, True)
    assert plan.info()["npairs"] == want_counts[1]
    for arr in C_NAMES:
        assert np.array_equal(plan.array(arr), want_arrays[arr]), arr
    plan2 = pkg.CPlan(ctx, gA, gB)                        # a new plan sees the new value
    plan2.spgemm()
    assert plan2.info()["npairs"] == op.npairs



@pytest.mark.parametrize("name", ["band_1500", "blockrows_10000", "blockrows_1600"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_step3_many_pair_kernel_matches_entry_per_lane(pkg, oracle, ctx, monkeypatch, name, dtype):
    """Deep plans send C tiles with 8 or more pairs to s3_band_kernel (records staged in LDS, one wave per tile) and the
    rest to the entry-per-lane kernel; PEM_S3_BAND=0 keeps everything in the latter.  Both must give the oracle's values
    bit for bit (same ascending-pair fma chain), and the band kernel must really have run."""
    rows, cols, I, J, V, tr = CASES[name]
    V = V.astype(dtype)
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, False, dtype=dtype)
    B = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True, dtype=dtype) if tr else A
    f32 = dtype == np.float32
    oa, ob = oracle.Csr(rows, cols, I, J, V.astype(np.float64), False), oracle.Csr(rows, cols, I, J, V.astype(np.float64), tr)
    orp, oci, ov = oracle.csr_spgemm(oa, ob, 1, f32=f32).arrays()
    got = {}
    for band in ("1", "0"):
        monkeypatch.setenv("PEM_S3_BAND", band)
        plan = pkg.CPlan(ctx, A, B)
        plan.spgemm()                                     # cold pass
        ctx.set_kernel_profiling(True)
        ctx.reset_kernel_stats()
        plan.spgemm()                                     # warm pass
        names = set(ctx.kernel_stats())
        ctx.set_kernel_profiling(False)
        info = plan.info()
        assert info["npairs"] >= 2 * info["ntiles_c"], "not a deep plan: the case does not reach the kernel under test"
        assert any(k.startswith("s3_band_kernel") for k in names) == (band == "1"), names
        rp, ci, v = plan.export_csr()
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
        assert v.dtype == dtype and np.array_equal(v.astype(np.float64), ov), f"{name} band={band}"
        got[band] = plan.array("c_vals").copy()
    assert np.array_equal(got["1"], got["0"])
