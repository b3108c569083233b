#!/usr/bin/env python3
"""Mean per-launch PMC values per kernel from rocprofv3 --pmc CSV output.  usage: pmc_summary.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per_dispatch = defaultdict(float)
    for r in csv.DictReader(open(f)):
        per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, c), v in per_dispatch.items():
        acc[k][c].append(v)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if flt in k:
        print(k[:110])
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
