"""Print the kernel timeline of the last full pass in a rocprofv3 --kernel-trace CSV (diagnostic)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 's1_reset' in r['Kernel_Name']]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
busy_end = 0
for r in rows[i0:i1]:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    print(f"{s / 1000:8.1f} {e / 1000:8.1f} {(e - s) / 1000:7.1f}  {r['Kernel_Name'][:70]}")
