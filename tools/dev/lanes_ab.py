"""Diagnostic: step-1 bin kernel times (each alone, kernel profiling) with the bins on four streams and on one: python tools/dev/lanes_ab.py [workload]"""
import importlib, os, sys
sys.path.insert(0, os.environ.get("PEM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pem-spgemm_amd"); standins = importlib.import_module("pem-spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M-r2"
m, n, I, J, V = standins.make(name, 1.0)
ctx = pkg.Context(0); A = pkg.Tiled.from_coo(ctx, m, n, I, J, V); plan = pkg.CPlan(ctx, A, A)
for ser in (0, 1):
    plan.set_option("s1_serial", ser)
    for _ in range(3): plan.spgemm()
    t = plan.timings() if hasattr(plan, "timings") else None
    ctx.set_kernel_profiling(True); ctx.reset_kernel_stats()
    for _ in range(3): plan.spgemm()
    ctx.set_kernel_profiling(False)
    print("s1_serial", ser, {k.replace("s1_rowsort_kernel", "rs"): round(v["total_ms"] / v["calls"] * 1e3, 1) for k, v in ctx.kernel_stats().items() if "rowsort" in k or "tiny" in k}, flush=True)
    import time
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); plan.spgemm(); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("   pass wall ms", [round(x, 3) for x in ts], flush=True)
