#!/bin/bash
# End-to-end run of the C++ CLI on full-size stand-ins written as .mtx (GPU box): parse + conversion + 10 timed passes + CSV.
set -e
OUT=${1:-gpurun_out/cli}
mkdir -p "$OUT"
python3 - "$OUT" <<'PY'
import sys, importlib, time
sys.path.insert(0, ".")
import __graft_entry__ as g
g.load_package(); st = importlib.import_module("pem_spgemm_amd.standins")
out = sys.argv[1]
import numpy as np
for name in ("scircuit", "webbase-1M", "mc2depi"):
    rows, cols, I, J, V = st.make(name)
    t = time.time()
    with open(f"{out}/{name}.mtx", "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% pem-spgemm_amd synthetic stand-in\n")
        f.write(f"{rows} {cols} {len(I)}\n")
        np.savetxt(f, np.column_stack([I + 1, J + 1, V]), fmt="%d %d %.17g")
    print(name, "written in", round(time.time() - t, 1), "s")
PY
cd "$OUT"
rm -f pemspgemm_benchmark_result.csv
for m in scircuit webbase-1M; do ../../pem-spgemm_amd/pemspgemm $PWD/$m.mtx 0 > $m.log 2>&1; tail -22 $m.log | head -18; done
../../pem-spgemm_amd/pemspgemm $PWD/mc2depi.mtx 0 1 > mc2depi.log 2>&1; tail -22 mc2depi.log | head -18
cat pemspgemm_benchmark_result.csv; echo
# tiled-format cache: cold run parses + converts + writes, warm run loads
mkdir -p cache
for m in scircuit webbase-1M; do
    for pass in cold warm; do
        PEM_CSV=cache.csv ../../pem-spgemm_amd/pemspgemm $PWD/$m.mtx 0 --cache $PWD/cache > cache_$m.$pass.log 2>&1
        echo "$m --cache $pass: $(grep -E 'tiled-format cache|total conversion overhead' cache_$m.$pass.log | tr '\n' ' ')"
    done
done
ls -l cache | awk '{print $5, $9}'
rm -rf cache *.mtx
