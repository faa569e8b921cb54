#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not k.startswith("void k_") and not k.startswith("k_"):
            continue
        agg[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for k in sorted(agg):
    print("kernel:", k)
    for c in sorted(agg[k]):
        per = defaultdict(float)
        for d, v in agg[k][c]:
            per[d] += v
        vals = list(per.values())
        print("  %-34s dispatches=%d mean=%.6g min=%.6g max=%.6g" % (c, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
