"""Do two half-matrix passes running on two streams at once finish sooner than one after the other?  (step 1 is VALU/LDS
bound, steps 2-3 are bound by the vector-memory path: out of phase they could share a CU)  python tools/overlap_probe.py"""
import importlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
standins = importlib.import_module("pem_spgemm_amd.standins")
rows, cols, I, J, V = standins.make("webbase-1M", 1.0)
c1, c2 = pkg.Context(0), pkg.Context(0)
A = pkg.Tiled.from_coo(c1, rows, cols, I, J, V)
b = pkg.split_tile_rows(c1, A, A, 2)
whole = pkg.CPlan(c1, A, A)
X = pkg.CPlan(c1, A, A, int(b[0]), int(b[1]))
Y = pkg.CPlan(c2, A, A, int(b[1]), int(b[2]))
N = 40
for p in (whole, X, Y):
    for _ in range(3):
        p.spgemm()


def run(p, n, delay=0.0):
    if delay:
        time.sleep(delay)
    for _ in range(n):
        p.spgemm()


t = time.perf_counter(); run(whole, N); t_whole = (time.perf_counter() - t) / N * 1e3
t = time.perf_counter(); run(X, N); tx = (time.perf_counter() - t) / N * 1e3
t = time.perf_counter(); run(Y, N); ty = (time.perf_counter() - t) / N * 1e3
for delay in (0.0, 0.0004, 0.0007):
    th = [threading.Thread(target=run, args=(X, N)), threading.Thread(target=run, args=(Y, N, delay))]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    tc = (time.perf_counter() - t - delay) / N * 1e3
    print(f"whole {t_whole:.3f} ms   halves one after the other {tx:.3f} + {ty:.3f} = {tx + ty:.3f} ms   both at once (second started {delay * 1e3:.1f} ms late) {tc:.3f} ms per pair")
