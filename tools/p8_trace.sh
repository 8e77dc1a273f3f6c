#!/bin/bash
# kernel timeline of ONE per-rank step at P ranks (tools/shard_sim.py --worlds P under rocprofv3 --kernel-trace; default P = 8): bash tools/p8_trace.sh [P]  (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_p8 -o p -- python tools/shard_sim.py --worlds ${1:-8} > gpurun_out/p8.log 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/prof_p8/p_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last complete step: find the last walk_sample kernel start, print from there to the end of the following slice_merge
idx=[i for i,r in enumerate(rows) if "walk_sample_kernel" in r["Kernel_Name"]]
# take the 5th from last step
s=idx[-5]; e=idx[-4]
t0=int(rows[s]["Start_Timestamp"]); prev=None
for r in rows[s:e]:
    s0,e0=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:60]
    print("%-60s start +%7.1f us dur %6.1f gap %5.1f" % (name,(s0-t0)/1e3,(e0-s0)/1e3,0 if prev is None else (s0-prev)/1e3))
    prev=e0
print("step span %.1f us" % ((int(rows[e]["Start_Timestamp"])-t0)/1e3))
PY
