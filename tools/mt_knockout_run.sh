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
PY
done
