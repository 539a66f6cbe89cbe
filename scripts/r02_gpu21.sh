#!/bin/bash
# round 2, GPU call 21: per-kernel times of the step through the module (autograd) vs the raw C-ABI sequence
O=gpurun_out/r2u; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for m in module cabi; do
  MODE=$m timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o run -- python $GRAFT_REPO_ROOT/scripts/host_overhead.py > $GRAFT_REPO_ROOT/$O/$m.txt 2> $GRAFT_REPO_ROOT/$O/$m.err
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py /tmp/prof_$m > $GRAFT_REPO_ROOT/$O/kernel_stats_$m.txt 2>&1
  cat $GRAFT_REPO_ROOT/$O/$m.txt; head -9 $GRAFT_REPO_ROOT/$O/kernel_stats_$m.txt | cut -c1-60,115-170
done
