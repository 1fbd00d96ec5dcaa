#!/bin/bash
# End-to-end runs of the C++ CLI on the seeded stand-ins (GPU box): generation (--standin) + conversion + 10 timed passes + CSV.
# usage: tools/cli_standins.sh <outdir>
set -e
OUT=${1:-gpurun_out/cli}
mkdir -p "$OUT"
CLI=$PWD/pem-spgemm_amd/pemspgemm
cd "$OUT"
rm -f pemspgemm_benchmark_result.csv
for m in scircuit webbase-1M; do $CLI --standin $m 0 > $m.log 2>&1; tail -22 $m.log | head -18; done
$CLI --standin mc2depi 0 1 > mc2depi.log 2>&1; tail -22 mc2depi.log | head -18
$CLI --standin cage15 0 > cage15.log 2>&1; tail -22 cage15.log | head -18
cat pemspgemm_benchmark_result.csv; echo
