#!/bin/bash
# kernel timeline of one replayed pass of one rank's share: tools/dev/share_trace.sh <workload> <nparts> <part>
export TMPDIR=/tmp
OUT=gpurun_out/share_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/slice_bench.py $1 $2 $3 > "$OUT/slice.txt" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
tail -1 "$OUT/slice.txt" | cut -c1-140
python3 tools/pass_timeline.py "$OUT/trace" ${4:-30} 2>&1 | head -30
find "$OUT/trace" -name "*.csv" -size +20M -delete
