"""Time one rank's share of an N-way row-block split on a single GPU (what each rank of `bench.py --gpus N` runs):
    python tools/slice_bench.py [workload] [nparts] [part] [passes]       (part -1: every part in turn)
Prints wall ms per pass as stream launches and as hipGraph replay (what bench.py times), and the step spans of the
stream passes; run under `rocprofv3 --kernel-trace` for the timeline."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
standins = importlib.import_module("pem_spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
part = int(sys.argv[3]) if len(sys.argv) > 3 else 0
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rows, cols, I, J, V = standins.make(name)
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
bounds = pkg.split_tile_rows(ctx, A, A, nparts)
for p in ([part] if part >= 0 else range(nparts)):
    plan = pkg.CPlan(ctx, A, A, int(bounds[p]), int(bounds[p + 1]))
    for _ in range(3):
        plan.spgemm()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(passes):
        plan.spgemm()
    ctx.synchronize()
    ms = (time.perf_counter() - t) * 1e3 / passes
    tm = ctx.timings()
    ctx.set_graph_replay(True)
    for _ in range(3):
        plan.spgemm()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(passes):
        plan.spgemm()
    ctx.synchronize()
    gms = (time.perf_counter() - t) * 1e3 / passes
    ctx.set_graph_replay(False)
    info = plan.info()
    print(f"{name} part {p}/{nparts}: tile rows [{bounds[p]}, {bounds[p + 1]})  {ms:.3f} ms/pass, {gms:.3f} as graph replay  step1 {tm['step1_ms']:.3f} step2 {tm['step2_ms']:.3f} "
          f"step3 {tm['step3_ms']:.3f}  pairs {info['npairs']} (all {info['npairs_all']}) C tiles {info['ntiles_c']} C nnz {info['nnz_c']}")
