#!/usr/bin/env python3
"""usage: pmc_calibrate_report.py <fetch_dir> <write_dir> <calibrate stdout json>: counter bytes per launch of every kernel of
tools/pmc_calibrate.py next to the pattern's known byte counts."""
import csv, glob, json, os, sys
from collections import defaultdict
fd, wd, js = sys.argv[1:4]
known = json.loads(open(js).read().strip().splitlines()[-1])


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r.get("Dispatch_Id", 0) or 0), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0))
    rows.sort()
    return rows


F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
print("tools/pmc_calibrate.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); counter bytes = KB x 1024 per launch")
print("known byte counts per launch:", json.dumps(known))
big = [(n, b) for _, n, b in F if b > 64e6]
bigw = {i: b for i, (_, n, b) in enumerate(W)}
print("\nlaunches with FETCH_SIZE > 64 MB, in dispatch order:")
for i, (_, n, b) in enumerate(F):
    if b > 64e6:
        wb = W[i][2] if i < len(W) and W[i][1] == n else float("nan")
        print(f"  {n[:90]:90s} FETCH {b / 1e9:9.3f} GB   WRITE {wb / 1e9:9.3f} GB")
