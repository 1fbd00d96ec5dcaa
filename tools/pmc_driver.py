"""a few repeat passes of one product, for rocprofv3 --pmc runs: python3 tools/pmc_driver.py [workload] [scale] [passes]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
standins = importlib.import_module("pem_spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows, cols, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
plan = pkg.CPlan(ctx, A, A)
for _ in range(1 + passes):
    plan.spgemm()
print("done", plan.info())
