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


def _run(n, extra):
    env = dict(os.environ, PEM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable]
    if n > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", "29571"]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_on_one_card_agree_with_one_rank():
    one = _run(1, ["--workload", "scircuit", "--scale", "0.25"])
    two = _run(2, ["--workload", "scircuit", "--scale", "0.25"])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "rowblock2+gather"
    for k in ("flop", "C_nnz", "C_tiles", "tile_pairs", "nnz"):
        assert one["config"][k] == two["config"][k], k
    assert two["value"] > 0 and two["scaling"] == "strong"
