#!/usr/bin/env python
"""Turn a rocprofv3 run directory (rocpd sqlite .db and/or *_kernel_stats.csv) into the short
text summary that is committed under profiles/ (kernel, calls, total us, avg us, %)."""
import csv
import glob
import os
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return [(short(r[0]), int(r[1]), float(r[2]), float(r[3]), float(r[4])) for r in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    d = sys.argv[1]
    rows = None
    for p in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = from_csv(p)
        break
    if rows is None:
        for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            rows = from_db(p)
            break
    if rows is None:
        raise SystemExit("no rocprofv3 output under " + d)
    print(f"{'kernel':112s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for name, calls, tot, avg, pct in rows[:25]:
        print(f"{name:112s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")


if __name__ == "__main__":
    main()
