#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc run directory: per kernel name, mean of every counter."""
import csv, glob, os, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"][:70]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in rows.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
