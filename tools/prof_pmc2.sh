#!/bin/bash
# Memory-path PMC profile of the hot kernels (TA / TCP / TCC / DRAM-vs-Infinity-Cache), counters in separate passes.
# usage: tools/prof_pmc2.sh <outdir> [driver args...]
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/pmc_driver.py "${ARGS[@]}" > "$OUT/$name.log" 2>&1
}
ARGS=("$@")
run p1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
run p2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_TAG_STALL_sum TCC_REQ_sum TD_TC_STALL_sum
run p3 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES
run p4 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR
python3 tools/pmc_table.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
