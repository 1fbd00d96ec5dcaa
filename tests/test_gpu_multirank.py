"""GPU: rehearse the N>1 bench path on ONE card (two ranks share cuda:0, gloo carries the gather).
The 8-GPU RCCL run is the driver's; this checks the row split, per-rank plans, device CSR export
and the gather code against the 1-rank result on real device memory."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra, launcher=True):
    """launcher=True: under torch.distributed.run, the driver's form; False: `python bench.py --gpus N` as typed, bench.py
    starts its own ranks as a child process"""
    env = dict(os.environ, PEM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable]
    if n > 1 and launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", "29571"]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _oracle_fingerprint(workload, scale, aat=False):
    """the fingerprint bench.py prints for the gathered C, computed from the CPU oracle's C of the same stand-in"""
    import importlib
    import numpy as np
    import __graft_entry__ as g
    g.load_package()
    standins = importlib.import_module("pem_spgemm_amd.standins")
    o = g.load_oracle()
    rows, cols, I, J, V = standins.make(workload, scale)
    a = o.Csr(rows, cols, I, J, V, False)
    b = o.Csr(rows, cols, I, J, V, True) if aat else a
    rp, ci, v = o.csr_spgemm(a, b, o.max_threads()).arrays()
    return dict(rows=len(rp) - 1, nnz=len(ci), rowptr_last=int(rp[-1]), colidx_sum=int(ci.astype(np.int64).sum()),
                rowptr_sum=int(rp.astype(np.int64).sum()), vals_sum=float(v.sum()), vals_abs_sum=float(np.abs(v).sum()))


def _same_matrix(got, want):
    """equal CSR arrays give equal integer sums; the value sums are taken in different orders (torch vs numpy)"""
    for k in ("rows", "nnz", "rowptr_last", "colidx_sum", "rowptr_sum"):
        assert got[k] == want[k], k
    assert abs(got["vals_sum"] - want["vals_sum"]) <= 1e-9 * want["vals_abs_sum"]
    assert abs(got["vals_abs_sum"] - want["vals_abs_sum"]) <= 1e-9 * want["vals_abs_sum"]


def test_two_ranks_on_one_card_agree_with_one_rank():
    one = _run(1, ["--workload", "scircuit", "--scale", "0.25"])
    two = _run(2, ["--workload", "scircuit", "--scale", "0.25"])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "rowblock2+gather"
    _same_matrix(two["exchange"]["gathered"], _oracle_fingerprint("scircuit", 0.25))     # the gathered C is the oracle's C
    for k in ("flop", "C_nnz", "C_tiles", "tile_pairs", "nnz"):
        assert one["config"][k] == two["config"][k], k
    assert two["value"] > 0 and two["scaling"] == "strong"
    # the same typed plainly: bench.py generates the input once, shares it and launches its own two ranks
    plain = _run(2, ["--workload", "scircuit", "--scale", "0.25"], launcher=False)
    assert plain["n_gpus"] == 2 and plain["exchange"]["gathered"] == two["exchange"]["gathered"]
    assert plain["exchange"]["ms_per_step"] > 0 and plain["exchange"]["bytes_to_root"] > 0


def test_two_ranks_a_at_row_blocks_tile_only_their_rows():
    """A*A^T on two ranks: B = A^T whole on both, each rank tiles only its own row block of A (cut from the COO with
    boundaries read off B's tile CSC); flop, sizes and the gathered C equal the one-rank run."""
    one = _run(1, ["--workload", "mc2depi", "--scale", "0.05"])
    two = _run(2, ["--workload", "mc2depi", "--scale", "0.05"], launcher=False)
    for k in ("flop", "C_nnz", "C_tiles", "tile_pairs", "nnz"):
        assert one["config"][k] == two["config"][k], k
    assert two["exchange"]["gathered"]["nnz"] == one["config"]["C_nnz"]
    _same_matrix(two["exchange"]["gathered"], _oracle_fingerprint("mc2depi", 0.05, aat=True))


def test_grid_partition_on_one_card_gathers_the_same_matrix():
    """4 ranks as a 2x2 (row block of A) x (column block of B) grid against 2 row-block ranks: same flop,
    same C, and the assembled CSR on the root has the same fingerprint (equal arrays give equal sums, value sums included)."""
    two = _run(2, ["--workload", "scircuit", "--scale", "0.25"])
    grid = _run(4, ["--workload", "scircuit", "--scale", "0.25", "--grid", "2x2"])
    assert grid["n_gpus"] == 4 and grid["config"]["parallelism"] == "grid2x2+gather"
    for k in ("flop", "C_nnz", "nnz"):
        assert two["config"][k] == grid["config"][k], k
    g2, g4 = two["exchange"]["gathered"], grid["exchange"]["gathered"]
    assert g2 == g4 and g2["nnz"] == two["config"]["C_nnz"] == g2["rowptr_last"]
    _same_matrix(g4, _oracle_fingerprint("scircuit", 0.25))
    aat = _run(3, ["--workload", "mc2depi", "--scale", "0.05", "--grid", "1x3"])          # A*A^T, B split only
    ref = _run(1, ["--workload", "mc2depi", "--scale", "0.05"])
    assert aat["config"]["flop"] == ref["config"]["flop"] and aat["config"]["C_nnz"] == ref["config"]["C_nnz"]
    assert aat["exchange"]["gathered"]["nnz"] == ref["config"]["C_nnz"]


def test_chunked_pipeline_gathers_the_same_matrix():
    """row blocks cut into chunks whose CSR travels while the next chunk computes -- same C on the root."""
    r = _run(2, ["--workload", "scircuit", "--scale", "0.25", "--chunks", "3"])
    pipe, g = r["exchange"]["pipelined"], r["exchange"]["gathered"]
    assert pipe["chunks"] == 3 and pipe["ms_per_step"] > 0
    fp = pipe["fingerprint"]
    assert (fp["nnz"], fp["colidx_sum"], fp["rowptr_sum"], fp["vals_sum"]) == (g["nnz"], g["colidx_sum"], g["rowptr_sum"], g["vals_sum"])
