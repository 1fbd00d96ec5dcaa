"""how long the driver takes for one device allocation of a given size (what a first pass pays per sizing phase):
    python tools/malloc_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
ctx = pkg.Context(0)
for mb in (1, 64, 256, 1024, 4096, 16384, 65536):
    ts = []
    for _ in range(3):
        ctx.synchronize()
        t = time.perf_counter()
        ctx.reserve(mb << 20)
        t1 = time.perf_counter()
        ctx.trim()
        t2 = time.perf_counter()
        ts.append(((t1 - t) * 1e3, (t2 - t1) * 1e3))
    print(f"{mb:6d} MiB: hipMalloc " + " ".join(f"{a:8.3f}" for a, _ in ts) + " ms   hipFree " + " ".join(f"{b:8.3f}" for _, b in ts) + " ms", flush=True)
