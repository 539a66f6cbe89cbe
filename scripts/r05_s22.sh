#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt; LAYER_KIND=tucker LAYER_REPS=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
python - <<'PY' > $O/tfno_timeline.txt
import csv, glob
rows = []
for p in glob.glob("/tmp/pt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_fft2d_fwd3" in r[2]][::2]
a, b = idx[-3], idx[-2]
t0 = rows[a][0]; busy = 0; prev = None
for s, e, n in rows[a:b]:
    gap = 0 if prev is None else (s - prev) / 1e3
    busy += e - s
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:6.1f}  {n[:70]}")
    prev = max(prev or e, e)
print(f"step: {(rows[b][0] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, {b - a} dispatches")
PY
cat $O/tfno_timeline.txt
