"""Diagnostic: step 3's kernel time on a stand-in (kernel profiling): python tools/s3_time.py [workload] [scale] [option=value ...]"""
import importlib, os, sys
sys.path.insert(0, os.environ.get("PEM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pem-spgemm_amd")
standins = importlib.import_module("pem-spgemm_amd.standins")
args = [a for a in sys.argv[1:] if "=" not in a]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
name = args[0] if args else "webbase-1M"
scale = float(args[1]) if len(args) > 1 else 1.0
m, n, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, m, n, I, J, V)
plan = pkg.CPlan(ctx, A, A)
for k, v in opts.items():
    plan.set_option(k, int(v))
for _ in range(3):
    plan.spgemm()
ctx.set_kernel_profiling(True)
ctx.reset_kernel_stats()
for _ in range(5):
    plan.spgemm()
ctx.set_kernel_profiling(False)
print("  ".join(f"{k} {v['total_ms'] / v['calls'] * 1e3:.1f} us" for k, v in ctx.kernel_stats().items() if k.startswith("s3_")))
