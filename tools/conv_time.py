"""Diagnostic: per-kernel times of the .mtx -> tiled conversion (COO already in HBM) on a stand-in: python tools/conv_time.py [workload] [scale]"""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("PEM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("pem-spgemm_amd")
standins = importlib.import_module("pem-spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
m, n, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0)
dI, dJ, dV = (torch.from_numpy(x).cuda() for x in (I, J, V))
torch.cuda.synchronize()
walls = []
for it in range(6):
    if it == 3:
        ctx.set_kernel_profiling(True)
        ctx.reset_kernel_stats()
    t0 = time.perf_counter()
    try:
        A = pkg.Tiled.from_coo_device(ctx, m, n, len(I), dI.data_ptr(), dJ.data_ptr(), dV.data_ptr())
        del A
    except pkg.PemError as e:      # (ablation builds sort wrongly: the kernels before the failure are still timed)
        print("conversion failed:", e)
    walls.append((time.perf_counter() - t0) * 1e3)
ctx.set_kernel_profiling(False)
print("wall ms (3 plain, 3 with kernel profiling):", " ".join(f"{w:.3f}" for w in walls))
tot = 0.0
for k, v in ctx.kernel_stats().items():
    per = v["total_ms"] / 3 * 1e3
    tot += per
    print(f"  {k:40s} {v['calls'] // 3:3d} launches  {per:8.1f} us per conversion")
print(f"  kernels in all {tot:.1f} us per conversion")
