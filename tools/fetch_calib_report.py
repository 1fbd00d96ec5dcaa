#!/usr/bin/env python3
"""Counters of tools/microbench/fetch_calib against its known byte counts: tools/fetch_calib_report.py <rocprofv3 dir> <program stdout>
FETCH_SIZE (KiB) as rocprofv3 derives it, and the bytes of the L2's fabric read requests counted by size
(32 * TCC_EA0_RDREQ_32B + 64 * TCC_EA0_RDREQ_64B + 128 * TCC_EA0_RDREQ_128B), whichever of them the run collected."""
import csv, glob, os, sys
d, out = sys.argv[1], sys.argv[2]
expect = {}
for row in csv.DictReader(open(out)):
    expect[row["kernel"]] = (int(row["expect_bytes"]), row["what"])
got = {}
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        got.setdefault(row["Kernel_Name"].split("(")[0], {})[row["Counter_Name"]] = float(row["Counter_Value"])
print(f"{'kernel':12s} {'expected bytes':>15s} {'FETCH_SIZE':>12s} {'ratio':>6s} {'by request size':>16s} {'ratio':>6s}   32B / 64B / 128B requests")
for k, (e, what) in expect.items():
    g = got.get(k, {})
    fs = g.get("FETCH_SIZE")
    n32, n64, n128 = g.get("TCC_EA0_RDREQ_32B_sum"), g.get("TCC_EA0_RDREQ_64B_sum"), g.get("TCC_EA0_RDREQ_128B_sum")
    by = None if None in (n32, n64, n128) else 32 * n32 + 64 * n64 + 128 * n128
    print(f"{k:12s} {e:15d} {'' if fs is None else int(fs * 1024):>12} {'' if fs is None else f'{fs * 1024 / e:6.3f}'} "
          f"{'' if by is None else int(by):>16} {'' if by is None else f'{by / e:6.3f}'}   "
          + ("" if by is None else f"{int(n32)} / {int(n64)} / {int(n128)}") + f"   {what}")
