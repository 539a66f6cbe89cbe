#!/bin/bash
# the per-rank step of an 8-rank group emulated on one device: kernel timeline of a TIMED step (early in the trace: the
# stage measurements that follow run the full-weight shapes of the one-rank layer)
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --stage-iters 2 --settle-ms 0"
rm -rf /tmp/p2; rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o run -- python $GRAFT_REPO_ROOT/bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --emulate-world 8 $Q > /dev/null 2>/tmp/p2.err
python - <<'PY' > $O/emu8_timeline.txt
import csv, glob
rows = []
for p in glob.glob("/tmp/p2/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_pl128_fwd" in r[2]][::2]
print("steps found:", len(idx))
for which in (12, len(idx) - 3):
    a, b = idx[which], idx[which + 1]
    t0 = rows[a][0]; busy = 0
    print(f"-- step {which}")
    for s, e, n in rows[a:b]:
        busy += e - s
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  {n[:100]}")
    print(f"step: {(rows[b][0] - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {b - a} dispatches")
PY
cat $O/emu8_timeline.txt
