"""print the headline fields of a bench.py JSON line: python tools/show_bench.py FILE"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"value {d['value']:.2f} {d['unit']}  ms_per_step {d['ms_per_step']:.4f}  steps_ms {d['steps_ms']['step1']:.3f}/{d['steps_ms']['step2']:.3f}/{d['steps_ms']['step3']:.3f}")
if d.get("t_total"):
    t = d["t_total"]
    print("t_total:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items() if k != "note"})
r = d.get("roofline") or {}
print("roofline:", {k: r.get(k) for k in ("kernel", "achieved", "frac", "frac_vs_measured_peak", "frac_kernel_bytes", "avg_launch_ms", "traffic")})
rp = d.get("roofline_pipeline") or {}
print("pipeline:", {k: rp.get(k) for k in ("frac", "frac_vs_measured_peak", "frac_kernel_spans", "t_kernel_ms")})
for k, v in d["kernels"].items():
    print(f"  {k:40s} {v['ms_per_step'] * 1e3:8.1f} us")
print("cpu:", d.get("cpu_baseline"))
