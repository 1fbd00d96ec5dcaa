"""wall + step spans of repeat passes (stream launches and graph replay): python tools/step_times.py [workload] [scale]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
standins = importlib.import_module("pem_spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows, cols, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
plan = pkg.CPlan(ctx, A, A)
for _ in range(4):
    plan.spgemm()
s = [0.0, 0.0, 0.0, 0.0]
n = 10
for _ in range(n):
    plan.spgemm()
    t = ctx.timings()
    for k, key in enumerate(("step1_ms", "step2_ms", "step3_ms", "spgemm_wall_ms")):
        s[k] += t[key] / n
ctx.set_graph_replay(True)
plan.spgemm()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    plan.spgemm()
ctx.synchronize()
gms = (time.perf_counter() - t0) * 1e3 / 20
print(f"{name}: stream step1 {s[0]:.3f} step2 {s[1]:.3f} step3 {s[2]:.3f} wall {s[3]:.3f} ms   graph {gms:.3f} ms  {os.environ.get('TAG', '')}")
