#!/bin/bash
# round 3, GPU call 24: TFNO step, final kernel stats + per-kernel ablations of this state
O=gpurun_out/r3za; mkdir -p $O
for i in 1 2; do (timeout 200 python scripts/tfno_time.py 2>&1 | tail -2) >> $O/tfno_time.txt; done; cat $O/tfno_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/tfno_kernel_stats_final.txt 2>&1
head -24 $O/tfno_kernel_stats_final.txt | cut -c1-170
TAG="full" timeout 120 python scripts/fmx_time.py 2>&1 | tail -3 >> $O/fmx_time.txt
TAG="valu (SC_FMX_OFF)" SC_FMX_OFF=1 timeout 120 python scripts/fmx_time.py 2>&1 | tail -3 >> $O/fmx_time.txt
TAG="mx" timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/fmx_time.txt
TAG="valu (SC_TK_VALU)" SC_TK_VALU=1 timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/fmx_time.txt
cat $O/fmx_time.txt
