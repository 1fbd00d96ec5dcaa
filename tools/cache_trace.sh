set -e
python3 - <<'PY'
import sys, importlib, time
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package(); st = importlib.import_module("pem_spgemm_amd.standins")
rows, cols, I, J, V = st.make("webbase-1M")
ctx = pkg.Context(0)
t=time.time(); A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V); print("from_coo", round((time.time()-t)*1e3,1), "ms")
t=time.time(); A.save("/tmp/w.pemtile"); print("save", round((time.time()-t)*1e3,1), "ms")
for i in range(3):
    t=time.time(); B = pkg.Tiled.load(ctx, "/tmp/w.pemtile"); print("load", round((time.time()-t)*1e3,1), "ms")
PY
