#!/bin/bash
# Round-end measurement set on the GPU box (one call): full GPU suite, the profile set of tools/prof_round.sh, and the bench
# lines of the other workloads of BASELINE.md section 5.   usage: tools/final_round.sh <outdir>
OUT=$1
mkdir -p "$OUT"
python -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.log" 2>&1 || { tail -20 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
bash tools/prof_round.sh "$OUT/prof" || exit 1
for w in webbase-1M-r2 scircuit mc2depi; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 1
done
python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_webbase_f32.json" 2> "$OUT/bench_f32.err" || exit 1
timeout -k 10 300 python bench.py --workload cage15 --steps 5 --warmup 2 --no-cpu-baseline --no-r2 > "$OUT/bench_cage15_one_gpu.json" 2> "$OUT/bench_cage15.err" || exit 1
python tools/slice_bench.py cage15 8 0 > "$OUT/slice_cage15_8way_part0.txt" 2>&1
echo done
