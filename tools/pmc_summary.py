#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per counter pass) into a per-kernel table.
Counter values are averaged per dispatch.  Fetched bytes come from the L2's fabric read requests counted by size
(32 n32 + 64 n64 + 128 n128: TCC_EA0_RDREQ_{32B,64B,128B}_sum); where a run collected FETCH_SIZE instead it is doubled --
profiles/r04_fetch_size_calibration.txt: on gfx950 every L2 miss is one 128-byte request, streaming or gather, and FETCH_SIZE
tallies 64 bytes for it (the guide's gfx950 correction, MI355X_MICROARCH.md HBM section, which that calibration extends from
wide streaming reads to 4-, 8- and 32-byte gathers).  WRITE_SIZE is reported as is, in bytes (the counters are in KiB)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def canon(name):
    """rocprofv3's kernel name -> the name the library's own per-kernel timing (PEM_LAUNCH_NAMED) and bench.py use"""
    base = name.split("(")[0].replace("pem::", "").replace("void ", "").strip()
    if base.startswith("s3_accumulate_wide_kernel<"):
        args = [a.strip() for a in base[base.index("<") + 1:base.rindex(">")].split(",")]
        flags = args[1:] + ["false"] * (6 - len(args))          # <VT, DEEP, BAND, DECODE, IDX32, MARK>
        tags = [t for t, f in zip(("deep", "band", "decode", "idx32", "mark"), flags) if f == "true"]
        return "s3_accumulate_wide_kernel<" + ",".join([args[0]] + tags) + ">"
    if base.startswith("s1_rowsort_kernel<"):
        args = [a.strip() for a in base[base.index("<") + 1:base.rindex(">")].split(",")]
        return "s1_rowsort_kernel<" + args[1] + (",rank" if len(args) > 5 and args[5] == "true" else "") + ">"
    return base

per = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values]
dur = defaultdict(list)
for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            short = canon(name)
            per[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for path in glob.glob(os.path.join(out, "sq", "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            short = canon(row["Kernel_Name"])
            dur[short].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
        "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]
print("kernel".ljust(34), "calls".rjust(5), "avg_us".rjust(9), " ".join(c.replace("SQ_", "").rjust(14) for c in cols))
rows = []
fetch_how = {}
for k, cs in per.items():
    n = max(len(v) for v in cs.values())
    avg = {c: (sum(cs[c]) / len(cs[c]) if cs.get(c) else float("nan")) for c in cols}
    n32, n64, n128 = (sum(cs[c]) / len(cs[c]) if cs.get(c) else None for c in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
    if None not in (n32, n64, n128):
        avg["FETCH_SIZE"] = 32 * n32 + 64 * n64 + 128 * n128          # bytes by request size
        fetch_how[k] = "request sizes"
    else:
        avg["FETCH_SIZE"] = avg["FETCH_SIZE"] * 1024 * 2              # KiB -> bytes, x2: 128-byte requests tallied at 64
        fetch_how[k] = "FETCH_SIZE x2"
    avg["WRITE_SIZE"] = avg["WRITE_SIZE"] * 1024
    d = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
    rows.append((d * len(dur.get(k, [])), k, n, d, avg))
for _, k, n, d, avg in sorted(rows, reverse=True):
    print(k[:34].ljust(34), str(len(dur.get(k, []))).rjust(5), f"{d:9.1f}", " ".join(f"{avg[c]:14.4g}" for c in cols))

# machine-readable traffic per launch (FETCH corrected x2 + WRITE), consumed by bench.py's roofline.traffic
import json
traffic = {}
for _, k, n, d, avg in rows:
    f, w = avg["FETCH_SIZE"], avg["WRITE_SIZE"]
    if f == f and w == w:
        traffic[k.strip()] = dict(hbm_bytes_per_launch=f + w, fetch_bytes=f, write_bytes=w, avg_us=d, fetch_from=fetch_how.get(k))
# which kernel sources the table belongs to: bench.py only quotes it for the same sources
import hashlib
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in ("primitives.hip", "convert.hip", "step1.hip", "step2.hip", "step3.hip", "export.hip", "spgemm.hip"):
    with open(os.path.join(_root, "pem-spgemm_amd", "csrc", _f), "rb") as _fh:
        _h.update(_fh.read())
traffic["__meta__"] = dict(kernels_sha=_h.hexdigest()[:16], source="rocprofv3 --pmc, separate passes: fetched bytes = 32/64/128-byte fabric read requests by size (or FETCH_SIZE x2), WRITE_SIZE; profiles/r04_fetch_size_calibration.txt")
with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1, sort_keys=True)
