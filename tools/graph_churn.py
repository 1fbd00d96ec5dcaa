"""Diagnostic for the hipGraphLaunch fault of round 3: rounds of create plan / warm passes under graph replay /
destroy plan.  PEM_DEBUG_GRAPH_DESTROY=1 makes the library destroy a plan's graph executable with the plan (the behaviour that
faulted); run under rocgdb to get the backtrace:  rocgdb -batch -ex run -ex bt --args python3 tools/graph_churn.py 40"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pem-spgemm_amd")
standins = importlib.import_module("pem-spgemm_amd.standins")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
name = sys.argv[2] if len(sys.argv) > 2 else "webbase-1M"
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
m, n, I, J, V = standins.make(name, scale)
ctx = pkg.Context(0)
ctx.set_graph_replay(True)
A = pkg.Tiled.from_coo(ctx, m, n, I, J, V)
mt = A.tile_rows
for r in range(rounds):
    lo = (r * 37) % max(mt // 2, 1)
    plan = pkg.CPlan(ctx, A, A, lo, min(mt, lo + mt // 2))
    for _ in range(4):
        plan.spgemm()
    info = plan.info()
    plan.close() if hasattr(plan, "close") else None
    del plan
    print(f"round {r}: pairs {info['npairs']} ok", flush=True)
print("done: no fault")
