#!/bin/bash
# Per-kernel PMC profile of one bench.py run on the GPU box (counters in their own passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage: tools/prof_pmc.sh <outdir> [bench args...]
set -e
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "${BENCH_ARGS[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err"
}
BENCH_ARGS=("$@")
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run fetch TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum   # fetched bytes by request size (profiles/r04_fetch_size_calibration.txt: FETCH_SIZE tallies half of them)
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
tail -n 60 "$OUT/summary.txt"
