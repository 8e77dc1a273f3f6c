#!/bin/bash
# per-kernel times of the device MT19937 generator per library variant (rocprof kernel trace of tools/mt_probe.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "base" ]; then unset PS_HIP_LIB; else export PS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ubench/_dbg/libps_mt$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mt_$v -o p -- python tools/mt_probe.py 23618800 > gpurun_out/mt_prof_$v.log 2>&1
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/prof_mt_$v/p_kernel_trace.csv")) if "mt_" in r["Kernel_Name"]]
last={}
for r in rows[-16:]:
    last.setdefault(r["Kernel_Name"].split("::")[-1].split("(")[0], []).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("variant $v:", {k: [round(x,1) for x in v] for k,v in last.items()})
t0=min(int(r["Start_Timestamp"]) for r in rows[-16:]); t1=max(int(r["End_Timestamp"]) for r in rows[-16:])
print("  last call: span %.1f us, kernel time %.1f us, gaps %.1f us" % ((t1-t0)/1e3, sum(sum(v) for v in last.values()), (t1-t0)/1e3-sum(sum(v) for v in last.values())))
prev=None
for r in sorted(rows[-16:], key=lambda r:int(r["Start_Timestamp"])):
    s0,e0=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-28s start +%7.1f us  dur %6.1f us  gap before %5.1f us" % (r["Kernel_Name"].split("::")[-1].split("(")[0][:28], (s0-t0)/1e3, (e0-s0)/1e3, 0 if prev is None else (s0-prev)/1e3))
    prev=e0
PY
done
