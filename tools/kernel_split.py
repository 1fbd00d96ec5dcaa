"""Per-kernel time split of a bench.py JSON line (stdin or file argument)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
k = d["kernels"]
tot = sum(v["ms_per_step"] for v in k.values())
for n, v in sorted(k.items(), key=lambda x: -x[1]["ms_per_step"]):
    print(f"{n:40s} {v['ms_per_step'] * 1e3:8.1f} us  {100 * v['ms_per_step'] / tot:5.1f} %")
print(f"sum of kernels {tot:.3f} ms; pass {d['ms_per_step']:.3f} ms ({d.get('launch', 'stream')})")
