"""Diagnostic: step-1 kernel times (kernel profiling, each kernel alone) on a stand-in: python tools/dev/k1_time.py [workload] [scale]"""
import importlib, os, sys
sys.path.insert(0, os.environ.get("PEM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pem-spgemm_amd"); standins = importlib.import_module("pem-spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
m, n, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0); A = pkg.Tiled.from_coo(ctx, m, n, I, J, V); plan = pkg.CPlan(ctx, A, A)
for _ in range(3): plan.spgemm()
ctx.set_kernel_profiling(True); ctx.reset_kernel_stats()
for _ in range(5): plan.spgemm()
ctx.set_kernel_profiling(False)
print(name, scale, {k.replace("s1_rowsort_kernel", "rs"): round(v["total_ms"] / v["calls"] * 1e3, 1) for k, v in ctx.kernel_stats().items() if k.startswith("s1_")}, flush=True)
