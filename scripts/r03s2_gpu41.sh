#!/bin/bash
O=gpurun_out/s2be; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_b && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o run -- python $GRAFT_REPO_ROOT/scripts/block_step_profile.py > $GRAFT_REPO_ROOT/$O/block.txt 2>&1)
python scripts/rocprof_summary.py /tmp/prof_b > $O/block_kernel_stats.txt 2>&1
tail -1 $O/block.txt; head -22 $O/block_kernel_stats.txt | cut -c1-190
