"""CPU: the N>1 path (tile-row split + gather of CSR slices) with gloo, world_size 2 and 3.
Slices are produced by the oracle here (no GPU in this container); the gather code is the one
bench.py runs over RCCL."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, bounds, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    from matgen import cases
    g.load_package()
    mg = importlib.import_module("pem_spgemm_amd.multigpu")
    o = g.load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols, I, J, V, tr = cases()["powerlaw_600"]
    A = o.Tiled(rows, cols, I, J, V)
    lo, hi = mg.slice_bounds(bounds, rank)
    rp, ci, v = o.Plan(A, A, lo, hi).export_csr()
    out = mg.gather_csr_slices(torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(v), dst=0)
    if rank == 0:
        full = o.Plan(A, A).export_csr()
        ok = all(np.array_equal(a.numpy(), b) for a, b in zip(out, full))
        q.put(bool(ok))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bounds", [(2, [0, 20, 38]), (3, [0, 0, 11, 38])])
def test_row_block_gather_equals_single_rank(world, bounds):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, bounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_2d(rank, world, port, ncb, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    from matgen import cases
    g.load_package()
    mg = importlib.import_module("pem_spgemm_amd.multigpu")
    o = g.load_oracle()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols, I, J, V, tr = cases()[case]
    BI, BJ = (J, I) if tr else (I, J)                      # B = A^T for the A*A^T cases
    brows, bcols = (cols, rows) if tr else (rows, cols)
    nrb = world // ncb
    rb = mg.balanced_tile_bounds(I, rows, nrb)
    cb = mg.balanced_tile_bounds(BJ, bcols, ncb)
    i, j = mg.grid_coords(rank, ncb)
    ma, mb = mg.restrict(I, rb[i], rb[i + 1]), mg.restrict(BJ, cb[j], cb[j + 1])
    A = o.Tiled(rows, cols, I[ma], J[ma], V[ma])           # only this rank's rows of A ...
    B = o.Tiled(brows, bcols, BI[mb], BJ[mb], V[mb])       # ... and columns of B; indices stay global
    rp, ci, v = o.Plan(A, B, rb[i], rb[i + 1]).export_csr()
    out = mg.gather_csr_blocks(torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(v), ncb, dst=0)
    if rank == 0:
        fa = o.Csr(rows, cols, I, J, V)
        fb = o.Csr(rows, cols, I, J, V, tr)
        full = o.csr_spgemm(fa, fb).arrays()
        ok = all(np.array_equal(a.numpy(), b) for a, b in zip(out, full))
        q.put(bool(ok))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ncb,case", [(4, 2, "powerlaw_600"), (3, 3, "rand_300"), (4, 2, "rect_70x40_AAt"), (2, 1, "rand_300")])
def test_2d_block_gather_equals_single_rank(world, ncb, case):
    """(row block of A) x (column block of B) grid; no rank ever holds all of A or of B."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world * 7 + ncb + (os.getpid() % 150)
    procs = [ctx.Process(target=_worker_2d, args=(r, world, port, ncb, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_concat_and_assemble_helpers_on_cpu():
    """the splicing used by the chunked pipeline and by the 2-D grid, against scipy slicing"""
    import scipy.sparse as sp
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    mg = importlib.import_module("pem_spgemm_amd.multigpu")
    M = sp.random(83, 117, 0.15, random_state=5, format="csr")
    M.sort_indices()

    def t(block, c0=0):
        block = block.tocsr()
        block.sort_indices()
        return (torch.from_numpy(block.indptr.astype(np.int32)), torch.from_numpy((block.indices + c0).astype(np.int32)),
                torch.from_numpy(block.data.copy()))
    cuts = [0, 16, 16, 48, 83]                                            # an empty slice in the middle
    rp, ci, v = mg.concat_csr_slices([t(M[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(rp.numpy(), M.indptr) and np.array_equal(ci.numpy(), M.indices) and np.array_equal(v.numpy(), M.data)
    rb, cb = [0, 2, 6], [0, 3, 3, 8]                                      # tile bounds; an empty column block
    blocks = []
    for i in range(2):
        for j in range(3):
            r0, r1, c0, c1 = 16 * rb[i], min(16 * rb[i + 1], 83), 16 * cb[j], min(16 * cb[j + 1], 117)
            blocks.append(t(M[r0:r1].tocsc()[:, c0:c1], c0))
    rp, ci, v = mg.assemble_csr_blocks(blocks, 3)
    assert np.array_equal(rp.numpy(), M.indptr) and np.array_equal(ci.numpy(), M.indices) and np.array_equal(v.numpy(), M.data)
    I = M.tocoo().row
    b = mg.balanced_tile_bounds(I, 83, 4)
    assert b[0] == 0 and b[-1] == 6 and all(x <= y for x, y in zip(b, b[1:]))
    assert mg.restrict(I, b[1], b[2]).sum() == ((I >= 16 * b[1]) & (I < 16 * b[2])).sum()


def test_recut_bounds_moves_rows_from_slow_ranks_to_fast_ones():
    """multigpu.recut_bounds: pure arithmetic behind the measured re-cut of the row split.  A synthetic cost -- a fixed part per
    rank, a part proportional to the weight, a penalty on three ranks -- is balanced by the first cut on weights only; re-cutting
    from the 'measured' times must keep the cut a partition of the rows, shrink the penalised ranks' blocks and lower the maximum."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    mg = importlib.import_module("pem_spgemm_amd.multigpu")
    rng = np.random.default_rng(0)
    w = rng.pareto(1.5, 60000) + 1.0
    pre = np.concatenate([[0.0], np.cumsum(w)])
    b = np.searchsorted(pre, pre[-1] * np.arange(9) / 8).astype(np.int32)
    b[0], b[-1] = 0, len(w)

    def truth(bb):
        t = 0.17 + 0.11 * (pre[bb[1:]] - pre[bb[:-1]]) / (pre[-1] / 8)
        t[[0, 1, 5]] += 0.03
        return t

    t0 = truth(b)
    b1 = mg.recut_bounds(w, b, t0, fixed=0.6 * t0.min())
    assert b1[0] == 0 and b1[-1] == len(w) and np.all(np.diff(b1) > 0) and b1.dtype == np.int32
    n0, n1 = np.diff(b), np.diff(b1)
    assert all(n1[p] < n0[p] for p in (0, 1, 5))
    t1 = truth(b1)
    assert t1.max() < t0.max() - 0.005
    b2 = mg.recut_bounds(w, b1, t1, fixed=0.5 * t1.min())
    assert truth(b2).max() <= t1.max() + 1e-9
    # equal times: nothing to move (up to one row at a boundary)
    same = mg.recut_bounds(w, b, np.full(8, 0.3))
    assert np.all(np.abs(same - b) <= 1)
    # one rank, and empty parts, are legal inputs
    assert list(mg.recut_bounds(w, [0, len(w)], [1.0])) == [0, len(w)]
    be = mg.recut_bounds(np.ones(10), [0, 0, 10], [0.0, 1.0])
    assert be[0] == 0 and be[-1] == 10
