#!/usr/bin/env python3
"""Timeline of one pass out of a rocprofv3 --kernel-trace CSV: start / end / duration (us, relative to the pass's first kernel),
hardware queue and kernel.  usage: tools/pass_timeline.py <dir or kernel_trace.csv> [pass index, default: the 12th]"""
import csv, glob, os, sys
src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(src)))
first = [i for i, k in enumerate(ks) if k[2].startswith("s1_expand")]
i0, i1 = first[which], first[which + 1] if which + 1 < len(first) else len(ks)
t0 = ks[i0][0]
print(f"# pass {which} of {len(first)} in {os.path.basename(src)}; times in us from the pass's first kernel")
print(f"{'start':>8s} {'end':>8s} {'dur':>7s}  queue kernel")
for k in ks[i0:i1]:
    name = k[2].split("(")[0].replace("void ", "")
    print(f"{(k[0] - t0) / 1e3:8.1f} {(k[1] - t0) / 1e3:8.1f} {(k[1] - k[0]) / 1e3:7.1f}  q{k[3]:<3s} {name[:90]}")
