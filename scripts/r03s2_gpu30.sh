#!/bin/bash
# round 3, session 2: kernel breakdown of the TFNO rank-0.1 step (current build)
O=gpurun_out/s2ai; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_t && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > $GRAFT_REPO_ROOT/$O/tfno_time.txt 2>&1)
python scripts/rocprof_summary.py /tmp/prof_t > $O/tfno_kernel_stats.txt 2>&1
cat $O/tfno_time.txt | tail -1; head -24 $O/tfno_kernel_stats.txt | cut -c1-175
