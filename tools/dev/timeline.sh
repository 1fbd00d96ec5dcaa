#!/bin/bash
# kernel timeline of one replayed pass of the headline config: tools/dev/timeline.sh <outfile>
export TMPDIR=/tmp
OUT=gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_traced.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 tools/pass_timeline.py "$OUT/trace" 30 > "$1" 2>&1; cat "$1"
find "$OUT/trace" -name "*.csv" -size +20M -delete
