#!/usr/bin/env python3
"""per-kernel averages of every counter found under <dir>/*/**/*counter_collection.csv (rocprofv3 --pmc passes)"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
per = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            short = row.get("Kernel_Name", "").split("(")[0].replace("pem::", "").replace("void ", "")
            per[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for path in glob.glob(os.path.join(out, "*", "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            short = row["Kernel_Name"].split("(")[0].replace("pem::", "").replace("void ", "")
            dur[short].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
names = sorted({c for k in per for c in per[k]})
rows = sorted(per, key=lambda k: -sum(dur.get(k, [0])))
for k in rows[:12]:
    d = dur.get(k, [])
    print(f"== {k}  calls/pass-set {len(d)}  avg {sum(d) / max(len(d), 1):.1f} us")
    for c in names:
        v = per[k].get(c)
        if v:
            print(f"     {c:38s} {sum(v) / len(v):16.4g}")
