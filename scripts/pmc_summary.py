#!/usr/bin/env python
"""Summarise rocprofv3 --pmc run directories: per kernel name, mean of every counter.
Reads the rocpd sqlite output (counters_collection view) and, for older runs, *counter_collection.csv."""
import collections
import csv
import glob
import os
import sqlite3
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                rows[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(p).cursor()
        try:
            for k, n, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
                rows[k[:70]][n].append(float(v))
        except sqlite3.Error:
            pass
for k, cs in rows.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
