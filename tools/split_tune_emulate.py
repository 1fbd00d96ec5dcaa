"""What multigpu.tune_row_bounds does on an N-GPU node, emulated on ONE GPU: the N row blocks are timed one after the other (each
alone on the card, as a rank on its own GPU would be), the cut is re-made from the times, and the new blocks are timed again.
    python tools/split_tune_emulate.py [workload] [nparts] [rounds]
Prints every round's per-part ms and the maximum -- the N-GPU pass time is the maximum."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package()
standins = importlib.import_module("pem_spgemm_amd.standins")
mg = importlib.import_module("pem_spgemm_amd.multigpu")
name = sys.argv[1] if len(sys.argv) > 1 else "webbase-1M"
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows, cols, I, J, V = standins.make(name)
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
w = pkg.tile_row_weights(ctx, A, A)
bounds = pkg.split_tile_rows(ctx, A, A, nparts)
ctx.set_graph_replay(True)


def measure(b, passes=20):
    out = []
    for p in range(nparts):
        plan = pkg.CPlan(ctx, A, A, int(b[p]), int(b[p + 1]))
        for _ in range(3):
            plan.spgemm()
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(passes):
            plan.spgemm()
        ctx.synchronize()
        out.append((time.perf_counter() - t) * 1e3 / passes)
        plan.close()
    return np.array(out)


best = None
for rnd in range(rounds + 1):
    t = measure(bounds)
    print(f"round {rnd}: max {t.max():.4f} mean {t.mean():.4f}  parts " + " ".join(f"{x:.3f}" for x in t) + "  rows " + " ".join(str(int(x)) for x in np.diff(bounds)))
    if best is None or t.max() < best[0]:
        best = (t.max(), bounds.copy())
    if rnd < rounds:
        bounds = mg.recut_bounds(w, bounds, t, fixed=(0.6 if rnd == 0 else 0.5) * float(t.min()))
print(f"best max {best[0]:.4f} ms")
