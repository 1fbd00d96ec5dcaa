export TMPDIR=/tmp
OUT=gpurun_out/r4r2
mkdir -p $OUT
cd /tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 bench.py --workload webbase-1M-r2 --steps 10 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_traced.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 tools/pass_timeline.py "$OUT/trace" 12 > "$OUT/timeline.txt" 2>&1; cat "$OUT/timeline.txt"
find "$OUT/trace" -name "*.csv" -size +20M -delete
