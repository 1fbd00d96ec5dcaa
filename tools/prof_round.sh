#!/bin/bash
# Round profile set on the GPU box: rocprofv3 kernel-trace stats of the default bench command, PMC passes (separate runs, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes), the memory-path table and the slice table.
# usage: tools/prof_round.sh <outdir>
OUT=$1
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" &&
bash tools/prof_pmc.sh "$OUT/pmc" --no-r2 > "$OUT/pmc.log" 2>&1 &&
python3 tools/slice_bench.py webbase-1M 8 -1 > "$OUT/slices_webbase_8way.txt" 2>&1 &&
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_webbase.json" 2> "$OUT/bench.err"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/rocprofv3_kernel_stats.csv"
