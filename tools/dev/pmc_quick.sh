#!/bin/bash
OUT=$1
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp && cd "$GRAFT_REPO_ROOT"
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/s3_time.py > "$OUT/$name.txt" 2> "$OUT/$name.err" || { grep -m3 -i "error\|exceeds" "$OUT/$name.err"; return 1; }
}
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE &&
run sq SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES &&
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for grp in ("ta1", "sq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "s3_" not in k: continue
            acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
    for k, d in acc.items():
        for c, v in sorted(d.items()): print(f"{grp} {c:36s} {v / n[(k, c)]:.4g}")
PY
