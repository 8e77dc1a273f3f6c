#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs by kernel: mean counter value per launch.
usage: pmc_summary.py <dir-or-csv> [<dir-or-csv> ...]"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k[:70])
    for c, (tot, n) in sorted(acc[k].items()):
        print(f"    {c:28s} launches={n:4d} mean={tot / n:16.1f}")
