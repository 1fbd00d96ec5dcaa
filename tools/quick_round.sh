#!/bin/bash
# Development loop on the GPU box (one call): parity tests of the hot path, the bench line, and a kernel-trace timeline of one pass.
# usage: tools/quick_round.sh <outdir> [pytest args]
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -q -x "$@" > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python tools/show_bench.py "$OUT/bench.json" | head -40
cd /tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_traced.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 tools/pass_timeline.py "$OUT/trace" 30 > "$OUT/timeline.txt" 2>&1; cat "$OUT/timeline.txt"
find "$OUT/trace" -name "*.csv" -size +20M -delete
