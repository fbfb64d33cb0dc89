#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (mean per dispatch)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0][-40:]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    if "nrh" not in k: continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g} max={max(v):.4g}")
