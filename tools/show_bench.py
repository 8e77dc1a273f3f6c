"""pretty-print the JSON line of bench.py (last line of the given file)"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.3e %s  ms/step %.4f  phases %s" % (d["value"], d["unit"], d["ms_per_step"], d["phase_ms"]))
print("workload:", d["config"]["workload"])
for k, v in d["kernels"].items():
    print("  %-28s x%.0f  avg %.4f ms  step %.4f ms  %8.1f %s  frac %.3f" % (k, v["launches_per_step"], v["avg_ms"], v["ms_per_step"], v["achieved"], v["unit"], v["frac"]))
print("roofline:", {k: d["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac")})
print("other mode:", d.get("other_rng_mode"))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu:", c["value"], c["unit"], c["cores"], "cores;", "x%.0f" % d["vs_cpu_baseline"])
