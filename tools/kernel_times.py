"""per-kernel device times of one product (or of one rank's row block of an N-way split):
    python tools/kernel_times.py [workload] [scale] [nparts] [part]   (env switches apply)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
standins = importlib.import_module("pem_spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows, cols, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
nparts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
part = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bounds = pkg.split_tile_rows(ctx, A, A, nparts)
plan = pkg.CPlan(ctx, A, A, int(bounds[part]), int(bounds[part + 1]))
for _ in range(3):
    try:
        plan.spgemm()
    except Exception as e:
        print("pass raised:", e)
ctx.set_kernel_profiling(True)
ctx.reset_kernel_stats()
n = 3
for _ in range(n):
    try:
        plan.spgemm()
    except Exception as e:
        print("pass raised:", e)
tot = 0.0
for k, v in ctx.kernel_stats().items():
    print(f"  {k:40s} {v['total_ms'] / n * 1e3:9.1f} us  x{v['calls'] / n:.0f}")
    tot += v["total_ms"] / n
print(f"  sum {tot * 1e3:.1f} us   timings {ctx.timings()}")
