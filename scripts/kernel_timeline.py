#!/usr/bin/env python
"""Timeline of the LAST step in a rocprofv3 --kernel-trace run directory: start / end of every dispatch relative to
the first one, the queue it ran on, and how long it overlapped the dispatch before it -- shows whether launches that
were issued to two streams actually ran side by side.  usage: kernel_timeline.py <dir> [n_last_dispatches]"""
import csv, glob, os, sqlite3, sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    break
if not rows:
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(p)
        cur = con.cursor()
        tabs = [t[0] for t in cur.execute("select name from sqlite_master where type in ('table','view')")]
        view = next((t for t in tabs if t == "kernels"), None)
        if view:
            cols = [c[1] for c in cur.execute(f"pragma table_info({view})")]
            q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
            for r in cur.execute(f"select start, end, {q}, name from {view}"):
                rows.append((int(r[0]), int(r[1]), str(r[2]), r[3]))
        break
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
prev_end = None
for s, e, q, name in rows:
    ov = 0 if prev_end is None else max(0, min(prev_end, e) - s)
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  queue {q:>4s}  overlap_prev {ov / 1e3:6.1f}  {name[:70]}")
    prev_end = e if prev_end is None else max(prev_end, e)
