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
