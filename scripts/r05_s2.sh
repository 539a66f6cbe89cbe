#!/bin/bash
# round 5, GPU session 2: the fused Tucker chain
O=gpurun_out/r05_s2; mkdir -p $O
python -m pytest tests/test_gpu_at_config.py -x -q -k "tucker or tfno" > $O/tests.log 2>&1; tail -5 $O/tests.log
python scripts/tkchain_time.py > $O/tkchain_time.txt 2>&1; cat $O/tkchain_time.txt
python scripts/tfno_time.py factorized > $O/tfno_fused.txt 2>&1; cat $O/tfno_fused.txt
SC_TKC_OFF=1 python scripts/tfno_time.py factorized > $O/tfno_nine.txt 2>&1; cat $O/tfno_nine.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python scripts/rocprof_summary.py /tmp/prof > $O/tfno_kernel_stats.txt 2>&1 || find /tmp/prof -name "*stats*" | head
head -40 $O/tfno_kernel_stats.txt
