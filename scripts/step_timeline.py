#!/usr/bin/env python
"""Timeline of ONE late step of a rocprofv3 --kernel-trace run (csv): every dispatch between the last two launches of
an anchor kernel (default k_pl128_fwd: the first kernel of a 3-D layer step), with the gap to the dispatch before it.
usage: step_timeline.py <dir> [anchor substring] [n steps back]"""
import csv, glob, os, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_pl128_fwd"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
# a step has two launches of a forward-type transform (forward + adjoint): take every second anchor
starts = idx[::2]
mid = len(starts) // 2 if back == 0 else len(starts) - back - 1   # back = 0: a step from the MIDDLE of the run
a, b = starts[mid], starts[mid + 1]
t0 = rows[a][0]
prev = None
busy = 0
for s, e, n in rows[a:b]:
    gap = 0 if prev is None else (s - prev) / 1e3
    busy += e - s
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n[:90]}")
    prev = max(prev or e, e)
print(f"step: {(rows[b][0] - t0) / 1e3:.1f} us from first dispatch to the next step's first, kernels busy {busy / 1e3:.1f} us, "
      f"{b - a} dispatches")
