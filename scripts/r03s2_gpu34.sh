#!/bin/bash
# round 3, session 2: kernel breakdown of a workload's step (rocprofv3 over scripts/graph_step_time.py's eager loop)
WL=${1:-fno2d_192_m64_c64_b32}
O=gpurun_out/s2an; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_t && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o run -- python $GRAFT_REPO_ROOT/scripts/graph_step_time.py $WL > $GRAFT_REPO_ROOT/$O/time_$WL.txt 2>&1)
python scripts/rocprof_summary.py /tmp/prof_t > $O/kernel_stats_$WL.txt 2>&1
tail -1 $O/time_$WL.txt; head -14 $O/kernel_stats_$WL.txt | cut -c1-200
