"""Diagnostic: per-chunk clocks of s1_expand_kernel (needs a library built with `make EXTRA=-DPEM_S1_DEBUG`)."""
import ctypes, importlib, sys, os
sys.path.insert(0, os.environ.get("PEM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(3):
    plan.spgemm()
dbg = (ctypes.c_ulonglong * (32768 * 8))()
lib.pem_debug_k1.restype = None
lib.pem_debug_k1.argtypes = [ctypes.c_void_p]
lib.pem_debug_k1(dbg)   # read once to find the accumulators' level before the measured pass
base_acc = np.frombuffer(dbg, dtype=np.uint64).reshape(32768, 8).astype(np.int64)[:, 4:].copy()
plan.spgemm()
buf = (ctypes.c_ulonglong * (32768 * 8))()
lib.pem_debug_k1.restype = None
lib.pem_debug_k1.argtypes = [ctypes.c_void_p]
lib.pem_debug_k1(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(32768, 8).astype(np.int64)
acc = a[:, 4:] - base_acc
sel = a[:, 0] > 0
a, acc = a[sel], acc[sel]
t0 = a[:, 0].min()
st, al, en, nw = (a[:, 0] - t0) / 100.0, (a[:, 1] - a[:, 0]) / 100.0, (a[:, 2] - a[:, 1]) / 100.0, a[:, 3]
print(f"chunks {len(a)}  kernel span {(a[:, 2].max() - t0) / 100.0:.1f} us")
print("start   pct 50/90/99/max:", np.percentile(st, [50, 90, 99, 100]).round(1))
print("alloc   pct 50/90/99/max:", np.percentile(al, [50, 90, 99, 100]).round(1))
print("loop    pct 50/90/99/max:", np.percentile(en, [50, 90, 99, 100]).round(1))
print("n_w     pct 50/90/99/max:", np.percentile(nw, [50, 90, 99, 100]).round(0))
it = np.ceil(nw / 256.0)
ok = it > 0
print("us per outer iteration (256 products): median %.2f  mean %.2f" % (np.median(en[ok] / it[ok]), en[ok].sum() / it[ok].sum()))
order = np.argsort(-(st + al + en))[:12]
for j in order:
    print(f"  chunk end {st[j] + al[j] + en[j]:7.1f} us: start {st[j]:6.1f} alloc {al[j]:5.1f} loop {en[j]:6.1f} n_w {nw[j]}")
its = np.maximum(it, 1)
for nm, col in (("walk", 0), ("gather", 1), ("chain wait", 2), ("stores", 3)):
    per = acc[:, col] / 100.0 / its
    print(f"{nm:10s} us per iteration (wave time): median {np.median(per):.2f}  mean {acc[:, col].sum() / 100.0 / its.sum():.2f}")
