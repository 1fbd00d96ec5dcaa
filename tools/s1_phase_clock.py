"""Diagnostic: phase clocks of the step-1 row-sort bins (needs a library built with `make EXTRA=-DPEM_S1_DEBUG`)."""
import ctypes, importlib, sys, os
sys.path.insert(0, os.environ.get("PEM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # PEM_PKG_ROOT: a copy built with -DPEM_S1_DEBUG
import numpy as np
pkg = importlib.import_module("pem-spgemm_amd")
standins = importlib.import_module("pem-spgemm_amd.standins")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
m, n, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, m, n, I, J, V)
plan = pkg.CPlan(ctx, A, A)
lib = pkg.lib()
lib.pem_debug_s1.restype = None
lib.pem_debug_s1.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 32)()
plan.spgemm()
plan.spgemm()
serial = os.environ.get("SERIAL", "1") == "1"
ctx.set_kernel_profiling(serial)      # serial: every kernel runs alone between two events
ctx.reset_kernel_stats()
lib.pem_debug_s1(buf, 1)
plan.spgemm()
lib.pem_debug_s1(buf, 0)
ms = {}
for k, v in ctx.kernel_stats().items():
    if "rowsort" in k:
        ms[int(k.split("<")[1].split(">")[0].split(",")[0])] = v["total_ms"] / v["calls"]
names = ["pieces", "load", "sort", "emit"]
for b, cap in enumerate([512, 2048, 8192, 32768]):
    v = [buf[b * 8 + k] for k in range(8)]
    rows = max(v[4], 1)
    tpu = 100.0   # wall_clock64 ticks per microsecond
    print(f"bin {cap:6d}: rows {v[4]:6d} kernel {ms.get(cap, 0) * 1e3:7.1f} us   " +
          "  ".join(f"{nm} {v[k] / rows / tpu:7.2f}" for k, nm in enumerate(names)) + f" us/row  slowest row {v[5] / tpu:7.1f} us")

if os.environ.get("BLOCKS"):
    blk = (ctypes.c_ulonglong * (4 * 1024 * 4))()
    lib.pem_debug_s1_blocks.restype = None
    lib.pem_debug_s1_blocks.argtypes = [ctypes.c_void_p]
    lib.pem_debug_s1_blocks(blk)
    a = np.frombuffer(blk, dtype=np.uint64).reshape(4, 1024, 4)
    for b, cap in enumerate([512, 2048, 8192, 32768]):
        nb = min(int(buf[b * 8 + 4]), 1024)
        if nb == 0:
            continue
        x = a[b, :nb]
        t0 = int(x[:, 0].min())
        print(f"== bin {cap}: {nb} blocks, first start -> last end {(int(x[:, 1].max()) - t0) / 100:.1f} us")
        hw = x[:, 2].astype(np.int64)
        cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = x[:, 3].astype(np.int64) & 15
        place = xcc * 1000 + se * 100 + sh * 16 + cu
        print("   distinct CUs used:", len(set(place.tolist())), " XCC histogram:", np.bincount(xcc, minlength=8).tolist())
        st = (x[:, 0].astype(np.int64) - t0) / 100.0
        en = (x[:, 1].astype(np.int64) - t0) / 100.0
        edges = np.arange(0, en.max() + 20, 20)
        print("   blocks started per 20 us:", np.histogram(st, edges)[0].tolist())
        print("   blocks running at t = 10, 30, ... us:", [int(((st <= t) & (en > t)).sum()) for t in edges[:-1] + 10])
        order = np.argsort(x[:, 0])
        for j in order[:: max(1, nb // 24)]:
            print(f"   block {j:4d} xcc {xcc[j]} se {se[j]} sh {sh[j]} cu {cu[j]:2d}  start {(int(x[j, 0]) - t0) / 100:7.1f}  dur {(int(x[j, 1]) - int(x[j, 0])) / 100:7.1f}")
