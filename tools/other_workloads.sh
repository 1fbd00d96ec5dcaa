#!/bin/bash
# bench lines of the other workloads of BASELINE.md section 5 (GPU box): usage tools/other_workloads.sh <outdir>
OUT=$1
mkdir -p "$OUT"
for w in webbase-1M-r2 scircuit mc2depi; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || { tail -5 "$OUT/bench_$w.err"; exit 1; }
  python tools/show_bench.py "$OUT/bench_$w.json" | head -1
done
python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-r2 > "$OUT/bench_webbase_f32.json" 2> "$OUT/bench_f32.err" || exit 1
python tools/show_bench.py "$OUT/bench_webbase_f32.json" | head -1
timeout -k 10 400 python bench.py --workload cage15 --steps 5 --warmup 2 --no-cpu-baseline --no-r2 > "$OUT/bench_cage15_one_gpu.json" 2> "$OUT/bench_cage15.err" || { tail -5 "$OUT/bench_cage15.err"; exit 1; }
python tools/show_bench.py "$OUT/bench_cage15_one_gpu.json" | head -14
python tools/slice_bench.py cage15 8 0 > "$OUT/slice_cage15_8way_part0.txt" 2>&1; tail -3 "$OUT/slice_cage15_8way_part0.txt"
