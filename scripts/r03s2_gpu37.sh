#!/bin/bash
O=gpurun_out/s2aq; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_t && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > $GRAFT_REPO_ROOT/$O/time.txt 2>&1)
find /tmp/prof_t -type f | head -5
python scripts/kernel_timeline.py /tmp/prof_t 26 2>&1 | tee $O/timeline.txt
