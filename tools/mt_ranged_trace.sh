#!/bin/bash
# kernel timeline of ONE ranged stream request (rank R of P, two layers' runs): bash tools/mt_ranged_trace.sh P R
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-8}; R=${2:-0}
cat > /tmp/mt_ranged.py <<PY
import os, sys
sys.path.insert(0, os.path.join("$GRAFT_REPO_ROOT", "movie-recommendation-engine_amd"))
import numpy as np, torch
from pinsage_hip import dense
n, P, rank = 23618800, $P, $R
h = n // 2
runs = [(r * h + rank * (h // P), r * h + (rank + 1) * (h // P)) for r in range(2)]
np.random.seed(0)
for _ in range(6):
    dense.mt19937_random_sample(n, "cuda", raw=True, advance=False, ranges=runs)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mt_ranged -o p -- python /tmp/mt_ranged.py > gpurun_out/mt_ranged.log 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/prof_mt_ranged/p_kernel_trace.csv")) if "mt_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
per=len(rows)//6
last=rows[-per:]
t0=int(last[0]["Start_Timestamp"]); prev=None
for r in last:
    s0,e0=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("   %-28s grid %-10s start +%7.1f us  dur %6.1f us  gap before %5.1f us" % (r["Kernel_Name"].split("::")[-1].split("(")[0][:28], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size","?"), (s0-t0)/1e3, (e0-s0)/1e3, 0 if prev is None else (s0-prev)/1e3))
    prev=e0
print("kernel time %.1f us" % sum((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in last))
PY
