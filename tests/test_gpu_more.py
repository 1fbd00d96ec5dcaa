"""GPU: boundary behaviour beyond the stage parity -- golden fixtures, CSR input, row slices,
error paths, the CLI end to end, and full-size properties."""
import glob
import importlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "spgemm_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[7:-4] for p in FIXTURES])
def test_gpu_matches_scipy_fixture(pkg, ctx, path):
    z = np.load(path, allow_pickle=False)
    rows, cols, tr = int(z["rows"]), int(z["cols"]), bool(z["transpose"])
    A = pkg.Tiled.from_coo(ctx, rows, cols, z["I"], z["J"], z["V"], False)
    B = pkg.Tiled.from_coo(ctx, rows, cols, z["I"], z["J"], z["V"], True) if tr else A
    plan = pkg.CPlan(ctx, A, B)
    plan.spgemm()
    rp, ci, v = plan.export_csr()
    assert np.array_equal(rp, z["c_rowptr"]) and np.array_equal(ci, z["c_colidx"])      # bit-exact structure
    np.testing.assert_allclose(v, z["c_vals"], rtol=1e-6, atol=1e-14)                    # north_star: 1e-6 relative fp64
    np.testing.assert_allclose(v, z["c_vals"], rtol=1e-12, atol=1e-14)                   # what the fma chain actually achieves


def test_from_csr_equals_from_coo(pkg, oracle, ctx):
    from matgen import cases
    rows, cols, I, J, V, _ = cases()["rand_300"]
    rp, ci, v = oracle.Csr(rows, cols, I, J, V).arrays()
    a, b = pkg.Tiled.from_csr(ctx, rows, cols, rp, ci, v), pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    for name in pkg.T_ARRAYS:
        assert np.array_equal(a.array(name), b.array(name)), name


def test_row_slices_concatenate_to_full(pkg, oracle, ctx):
    from matgen import cases
    rows, cols, I, J, V, _ = cases()["powerlaw_600"]
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    full = pkg.CPlan(ctx, A, A)
    full.spgemm()
    frp, fci, fv = full.export_csr()
    bounds = pkg.split_tile_rows(ctx, A, A, 4)
    assert bounds[0] == 0 and bounds[-1] == A.tile_rows and np.all(np.diff(bounds) >= 0)
    parts, oparts = [], []
    oA = oracle.Tiled(rows, cols, I, J, V)
    for g in range(4):
        p = pkg.CPlan(ctx, A, A, int(bounds[g]), int(bounds[g + 1]))
        p.spgemm()
        parts.append(p.export_csr())
        op = oracle.Plan(oA, oA, int(bounds[g]), int(bounds[g + 1]))
        from prune_ref import expected
        want, _ = expected(op, oA, oA)
        for name in ("c_tile_rowptr", "c_tile_rowidx", "c_tile_colidx", "pairs_offset", "pairs_a", "pairs_b", "c_mask", "c_vals"):
            assert np.array_equal(p.array(name), want[name]), (g, name)
    assert np.array_equal(np.concatenate([p[1] for p in parts]), fci)
    assert np.array_equal(np.concatenate([p[2] for p in parts]), fv)
    offs = np.cumsum([0] + [len(p[1]) for p in parts[:-1]])
    assert np.array_equal(np.concatenate([[0]] + [p[0][1:] + o for p, o in zip(parts, offs)]), frp)


def test_error_paths(pkg, ctx):
    I = np.array([0, 0, 1], np.int32)
    with pytest.raises(pkg.PemError) as e:
        pkg.Tiled.from_coo(ctx, 4, 4, I, I, np.ones(3))
    assert e.value.status == -2                      # PEM_E_DUPLICATE
    with pytest.raises(pkg.PemError) as e:
        pkg.Tiled.from_coo(ctx, 4, 4, np.array([5], np.int32), np.array([0], np.int32), np.ones(1))
    assert e.value.status == -1                      # PEM_E_INVALID: index out of range
    A = pkg.Tiled.from_coo(ctx, 4, 6, np.array([1], np.int32), np.array([2], np.int32), np.ones(1))
    with pytest.raises(pkg.PemError) as e:
        pkg.CPlan(ctx, A, A)                         # 4x6 times 4x6
    assert e.value.status == -1
    B = pkg.Tiled.from_coo(ctx, 4, 6, np.array([1], np.int32), np.array([2], np.int32), np.ones(1), True)
    p = pkg.CPlan(ctx, A, B)
    with pytest.raises(pkg.PemError) as e:
        p.step2()
    assert e.value.status == -6                      # PEM_E_STATE
    with pytest.raises(pkg.PemError):
        p.export_csr()
    p.spgemm()
    assert p.info()["nnz_c"] == 1


def test_set_option_sends_the_stepwise_api_back_to_the_last_untouched_step(pkg, oracle, ctx):
    """pem_cplan_set_option (include/pem_spgemm.h): a step that ran under the old value does not feed a step that runs under the
    new one -- the step-wise calls fail with PEM_E_STATE until the plan has been taken through the steps the option touches."""
    import matgen
    rows, cols, I, J, V, _ = matgen.cases()["powerlaw_600"]
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    want = oracle.Plan(oracle.Tiled(rows, cols, I, J, V), oracle.Tiled(rows, cols, I, J, V)).export_csr()
    p = pkg.CPlan(ctx, A, A)
    p.spgemm()
    for opt, value, first_ok in (("s3_mark", 0, 3), ("s3_decode", 0, 2), ("prune", 0, 1)):
        p.set_option(opt, value)
        with pytest.raises(pkg.PemError) as e:
            p.export_csr()                                   # step 3 has not run under the new value
        assert e.value.status == -6, opt                     # PEM_E_STATE
        steps = (p.step1, p.step2, p.step3)
        for k in (3, 2):                                     # a later step than the first one allowed: refused, nothing run
            if k > first_ok:
                with pytest.raises(pkg.PemError) as e:
                    steps[k - 1]()
                assert e.value.status == -6, (opt, k)
        for k in range(first_ok, 4):
            steps[k - 1]()
        got = p.export_csr()
        for g, w in zip(got, want):
            assert np.array_equal(g, w), opt


def test_refused_kernel_launch_is_reported(pkg, ctx):
    """PEM_LAUNCH (csrc/pem_internal.h): a launch the runtime refuses does not pass silently -- the next synchronising call returns
    PEM_E_HIP and names the kernel; after the report the context works again."""
    ctx.synchronize()
    ctx.debug_refused_launch()                           # asynchronous: PEM_OK
    with pytest.raises(pkg.PemError) as e:
        ctx.synchronize()
    assert e.value.status == -5 and "dbg_noop_kernel" in str(e.value)
    ctx.synchronize()                                    # reported once
    A = pkg.Tiled.from_coo(ctx, 4, 4, np.array([1], np.int32), np.array([2], np.int32), np.ones(1))
    assert A.ntiles == 1


def test_cli_end_to_end(pkg, oracle, standins, tmp_path):
    hostio = importlib.import_module("pem_spgemm_amd.hostio")
    rows, cols, I, J, V = standins.make("scircuit", scale=0.01)
    mtx = str(tmp_path / "mini.mtx")
    standins.write_mtx(mtx, rows, cols, I, J, V)
    env = dict(os.environ, PEM_RESULT_DIR=str(tmp_path), PEM_CSV=str(tmp_path / "r.csv"), PEM_REPEAT="2")
    out = subprocess.run([hostio.CLI_PATH, mtx, "1"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "<---Program done--->" in out.stdout and "GFlops" in out.stdout
    a = oracle.Csr(rows, cols, I, J, V)
    rp, ci, v = oracle.csr_spgemm(a, a).arrays()
    assert int((tmp_path / "SPGEMM_RESULT_NNZ.txt").read_text()) == len(ci)
    r_file = np.loadtxt(tmp_path / "SPGEMM_RESULT_ROWS.txt", dtype=np.int64, ndmin=1)
    c_file = np.loadtxt(tmp_path / "SPGEMM_RESULT_COLS.txt", dtype=np.int64, ndmin=1)
    v_file = np.loadtxt(tmp_path / "SPGEMM_RESULT_VALS.txt", dtype=np.float64, ndmin=1)
    assert np.array_equal(r_file, np.repeat(np.arange(rows), np.diff(rp))) and np.array_equal(c_file, ci)
    np.testing.assert_allclose(v_file, v, rtol=0, atol=5e-18 + 1e-17)     # '%.17f' text round trip
    rec = (tmp_path / "r.csv").read_text().split("\n")[1].split(",")
    assert rec[0] == "mini" and int(rec[2]) == len(ci) and len(rec) >= 14
    # A*A^T through the third argument; rectangular without it is refused (spgemm.cu:782-786)
    out = subprocess.run([hostio.CLI_PATH, mtx, "0", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "Not saving results" in out.stdout


@pytest.mark.parametrize("name,scale", [("scircuit", 1.0), ("mc2depi", 1.0), ("webbase-1M", 1.0)])
def test_full_size_against_cpu_port(pkg, oracle, standins, ctx, name, scale):
    """BASELINE sizes: whole C bit-exact against the OpenMP Gustavson port (seconds on the box's
    host cores) + size-independent properties (checksum identity, sortedness, idempotent re-run)."""
    rows, cols, I, J, V = standins.make(name, scale)
    tr = name == "mc2depi"
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    B = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True) if tr else A
    plan = pkg.CPlan(ctx, A, B)
    plan.spgemm()
    rp, ci, v = plan.export_csr()
    # sortedness / structure sanity
    assert rp[0] == 0 and rp[-1] == len(ci) and np.all(np.diff(rp) >= 0)
    inner = np.ones(len(ci), bool)
    inner[rp[:-1][np.diff(rp) > 0]] = False
    assert np.all(np.diff(ci.astype(np.int64))[inner[1:]] > 0), "columns must ascend inside every row"
    # checksum identity: sum(C) = (1^T A)(B 1) in exact-ish arithmetic (fp64 tolerance on ~1e8 terms)
    colsum_a = np.bincount(J, weights=V, minlength=cols)
    rowsum_b = np.bincount(J if tr else I, weights=V, minlength=cols if tr else rows)
    scale_abs = np.abs(colsum_a) @ np.abs(rowsum_b) + 1.0
    assert abs(v.sum() - colsum_a @ rowsum_b) <= 1e-9 * scale_abs
    # flop and sizes against the CPU port, then the whole result bit for bit
    oa = oracle.Csr(rows, cols, I, J, V)
    ob = oracle.Csr(rows, cols, I, J, V, True) if tr else oa
    rp0, ci0, v0 = oracle.csr_spgemm(oa, ob, oracle.max_threads()).arrays()
    assert np.array_equal(rp, rp0) and np.array_equal(ci, ci0)
    assert np.array_equal(v, v0)
    plan.spgemm()
    rp2, ci2, v2 = plan.export_csr()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(v2, v)


def test_cage15_rank_slice_against_cpu_port(pkg, oracle, standins, ctx):
    """BASELINE configs[4]: rank 0's share of the 8-way row-block split of the FULL-SIZE cage15 stand-in (5 154 859^2,
    99 199 551 nnz).  B has 322 179 tile columns, so step 1 runs its 64-bit-key row sorts, and B is far larger than
    the caches.  The exported CSR of the slice must equal the OpenMP Gustavson port on the same rows bit for bit."""
    rows, cols, I, J, V = standins.make("cage15", 1.0)
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    assert A.tile_cols > (1 << 17)
    bounds = pkg.split_tile_rows(ctx, A, A, 8)
    lo, hi = int(bounds[0]), int(bounds[1])
    plan = pkg.CPlan(ctx, A, A, lo, hi)
    plan.spgemm()
    plan.spgemm()                                                  # warm pass: same result
    rp, ci, v = plan.export_csr()
    info = plan.info()
    assert info["row_begin"] == 16 * lo and info["nnz_c"] == len(ci)
    r0, r1 = info["row_begin"], info["row_end"]
    sel = (I >= r0) & (I < r1)
    oa = oracle.Csr(r1 - r0, cols, I[sel] - r0, J[sel], V[sel])     # the slice's rows of A ...
    ob = oracle.Csr(rows, cols, I, J, V)                            # ... times all of B
    del I, J, V, sel
    rp0, ci0, v0 = oracle.csr_spgemm(oa, ob, oracle.max_threads()).arrays()
    assert np.array_equal(rp, rp0) and np.array_equal(ci, ci0)
    assert np.array_equal(v, v0)


def test_cage15_whole_matrix_in_eight_slices_against_cpu_port(pkg, oracle, standins, ctx):
    """BASELINE configs[4], the whole of it: all eight row blocks of the full-size cage15 stand-in, one after the other on this
    GPU, each compared with the OpenMP Gustavson port on the same rows -- entry counts, row-pointer and column-index sums
    exactly, value sums to 1e-9 of the absolute sum (round 2 compared rank 0's block only; that block is still compared
    array by array above).  Together the blocks are the whole C: the sizes must add up to the product's literature size
    (~0.93 G entries, the stand-in is calibrated to within 5 %)."""
    rows, cols, I, J, V = standins.make("cage15", 1.0)
    A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
    bounds = pkg.split_tile_rows(ctx, A, A, 8)
    ob = oracle.Csr(rows, cols, I, J, V)
    total_nnz, total_flop = 0, pkg.flop_count(ctx, A, A)
    for part in range(8):
        lo, hi = int(bounds[part]), int(bounds[part + 1])
        plan = pkg.CPlan(ctx, A, A, lo, hi)
        plan.spgemm()
        rp, ci, v = plan.export_csr()
        info = plan.info()
        plan.close()
        r0, r1 = info["row_begin"], info["row_end"]
        sel = (I >= r0) & (I < r1)
        oa = oracle.Csr(r1 - r0, cols, I[sel] - r0, J[sel], V[sel])
        rp0, ci0, v0 = oracle.csr_spgemm(oa, ob, oracle.max_threads()).arrays()
        assert len(ci) == len(ci0) == info["nnz_c"], part
        assert int(rp.astype(np.int64).sum()) == int(rp0.astype(np.int64).sum()), part
        assert int(ci.astype(np.int64).sum()) == int(ci0.astype(np.int64).sum()), part
        assert abs(float(v.sum()) - float(v0.sum())) <= 1e-9 * float(np.abs(v0).sum()), part
        assert np.array_equal(rp, rp0), part                   # (cheap: one int per row)
        total_nnz += len(ci)
        del rp, ci, v, rp0, ci0, v0, oa, sel
    assert abs(total_nnz / 0.93e9 - 1.0) < 0.05 and abs(total_flop / 2.08e9 - 1.0) < 0.05, (total_nnz, total_flop)


def test_cli_gpus_path_gathers_through_rccl_library(pkg, oracle, standins, tmp_path):
    """`pemspgemm --gpus N` (one context per device, slices gathered by libpemmgpu.so over RCCL): with the one GPU of this
    box the communicator has one rank, but conversion per rank, the row split, the threaded passes, the device CSR export
    and the assembly all run -- the result files must equal the oracle's C, like the single-device path's."""
    import scipy.io
    hostio = importlib.import_module("pem_spgemm_amd.hostio")
    rows, cols, I, J, V = standins.make("scircuit", scale=0.01)
    mtx, fc = str(tmp_path / "mini.mtx"), str(tmp_path / "C.mtx")
    standins.write_mtx(mtx, rows, cols, I, J, V)
    env = dict(os.environ, PEM_RESULT_DIR=str(tmp_path), PEM_CSV=str(tmp_path / "r.csv"), PEM_REPEAT="2")
    out = subprocess.run([hostio.CLI_PATH, mtx, "1", "--gpus", "1", "--out", fc], env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "gather of C to GPU 0 over RCCL" in out.stdout and "1 GPUs" in out.stdout
    a = oracle.Csr(rows, cols, I, J, V)
    rp, ci, v = oracle.csr_spgemm(a, a).arrays()
    assert int((tmp_path / "SPGEMM_RESULT_NNZ.txt").read_text()) == len(ci)
    assert np.array_equal(np.loadtxt(tmp_path / "SPGEMM_RESULT_COLS.txt", dtype=np.int64, ndmin=1), ci)
    assert np.array_equal(np.loadtxt(tmp_path / "SPGEMM_RESULT_ROWS.txt", dtype=np.int64, ndmin=1), np.repeat(np.arange(rows), np.diff(rp)))
    C = scipy.io.mmread(fc).tocsr()
    C.sort_indices()
    assert np.array_equal(C.indptr, rp) and np.array_equal(C.indices, ci) and np.array_equal(C.data, v)
    rec = (tmp_path / "r.csv").read_text().split("\n")[1].split(",")
    assert rec[14] == "1" and float(rec[20]) > 0 and len(rec) == 28          # gpus, first-pass ms, gather ms, chunked pass, split tuning
    assert int(rec[22]) >= 2 and float(rec[23]) > 0 and "chunks per rank, gather overlapped" in out.stdout     # the chunked pass ran
    # ... and the chunked pass itself through the C ABI: three chunks of the one rank, assembled CSR = the oracle's
    crp, cci, cv, pass_ms, tail_ms = hostio.mgpu_chunked_pass([0], rows, cols, I, J, V, 3)
    assert np.array_equal(crp, rp) and np.array_equal(cci, ci) and np.array_equal(cv, v) and pass_ms > 0
    # a device that does not exist is refused, not ignored
    out = subprocess.run([hostio.CLI_PATH, mtx, "0", "--gpus", "64"], env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 2 and "visible" in out.stderr


def test_cli_distinct_b_and_mtx_output(pkg, oracle, standins, tmp_path):
    """C = A*B with two files and a Matrix-Market result (beyond the reference's A^2 / A*A^T)."""
    import scipy.io
    hostio = importlib.import_module("pem_spgemm_amd.hostio")
    rng = np.random.default_rng(3)
    m, k, n = 70, 45, 90

    def rand(r, c, nz):
        key = rng.choice(r * c, nz, replace=False)
        return (key // c).astype(np.int32), (key % c).astype(np.int32), rng.uniform(-1, 1, nz)

    AI, AJ, AV = rand(m, k, 400)
    BI, BJ, BV = rand(k, n, 380)
    fa, fb, fc = str(tmp_path / "A.mtx"), str(tmp_path / "B.mtx"), str(tmp_path / "C.mtx")
    standins.write_mtx(fa, m, k, AI, AJ, AV)
    standins.write_mtx(fb, k, n, BI, BJ, BV)
    env = dict(os.environ, PEM_CSV=str(tmp_path / "r.csv"), PEM_REPEAT="1")
    out = subprocess.run([hostio.CLI_PATH, fa, "0", "--B", fb, "--out", fc], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    rp, ci, v = oracle.csr_spgemm(oracle.Csr(m, k, AI, AJ, AV), oracle.Csr(k, n, BI, BJ, BV)).arrays()
    C = scipy.io.mmread(fc).tocsr()
    C.sort_indices()
    assert C.shape == (m, n) and np.array_equal(C.indptr, rp) and np.array_equal(C.indices, ci) and np.array_equal(C.data, v)
    # shape mismatch is refused
    out = subprocess.run([hostio.CLI_PATH, fa, "0", "--B", fa], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "inner dimensions differ" in out.stdout


def test_graph_replay_is_bit_identical_and_survives_reallocation(pkg, oracle, ctx):
    """pem_set_graph_replay: repeat passes run as one captured hipGraph.  Same arrays as plain launches; the graph is
    re-captured when buffers were reallocated in between (an export, another plan growing the context's scratch)."""
    from matgen import cases
    from prune_ref import expected
    names = ["powerlaw_600", "blockrows_1600", "rect_70x40_AAt", "hub_row_4000"]
    plans, wants = [], []
    for name in names:
        rows, cols, I, J, V, tr = cases()[name]
        A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
        B = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V, True) if tr else A
        oA = oracle.Tiled(rows, cols, I, J, V)
        oB = oracle.Tiled(rows, cols, I, J, V, True) if tr else oA
        p = pkg.CPlan(ctx, A, B)
        p.spgemm()
        plans.append((p, A, B))
        wants.append(expected(oracle.Plan(oA, oB), oA, oB)[0])
    ctx.set_graph_replay(True)
    try:
        for rnd in range(3):                                  # interleaved plans: each keeps its own graph
            for (p, _, _), want in zip(plans, wants):
                p.spgemm()
                t = ctx.timings()
                assert t["step1_ms"] == 0.0 and t["spgemm_wall_ms"] > 0.0     # replayed: no step split
                for arr in ("pairs_a", "pairs_b", "c_mask", "c_tile_nnz_ptr", "c_rowcolidx", "c_vals", "c_tile_rowidx", "c_rowptr"):
                    assert np.array_equal(p.array(arr), want[arr]), (rnd, arr)
                if rnd == 1:
                    p.export_csr()                            # grows export scratch -> next replay must re-capture
        # a much larger plan on the same context grows the shared scan / sort scratch under the old graphs
        rows, cols, I, J, V, _ = cases()["blockrows_10000"]
        big = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
        pb = pkg.CPlan(ctx, big, big)
        pb.spgemm()
        pb.spgemm()
        for (p, _, _), want in zip(plans, wants):
            p.spgemm()
            assert np.array_equal(p.array("c_vals"), want["c_vals"])
    finally:
        ctx.set_graph_replay(False)
    plans[0][0].spgemm()
    assert ctx.timings()["step1_ms"] > 0.0                    # replay off: step split is back
