#!/bin/bash
# Memory-path counters (texture addresser / vector L1) of step 3's kernel on the webbase-1M stand-in, one pass per counter group.
# usage: tools/s3_mempath_pmc.sh <outdir>
OUT=$1
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp && cd "$GRAFT_REPO_ROOT"
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/s3_time.py > "$OUT/$name.txt" 2> "$OUT/$name.err" || { grep -m3 -i "error\|exceeds" "$OUT/$name.err"; return 1; }
}
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE &&
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE &&
run tcp1 TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE &&
run tcp2 TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE &&
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for grp in ("ta1", "ta2", "tcp1", "tcp2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not k.startswith("void s3_") and "s3_" not in k: continue
            acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k[:60], r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(grp, k)
        for c, v in sorted(d.items()):
            print(f"   {c:42s} {v / n[(k, c)]:.4g} per launch")
PY
