#!/bin/bash
# round 2, GPU call 17: Tucker mode-factor kernels after register tiling (tests, TFNO step time, kernel stats)
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tucker or tfno or factor" 2>&1 | grep -E "passed|failed|Error" | tail -5) > $O/pytest.log
cat $O/pytest.log
(timeout 200 python scripts/tfno_time.py 2>&1 | tail -2) > $O/tfno_time.txt
cat $O/tfno_time.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/tfno_kernel_stats.txt 2>&1
head -16 $O/tfno_kernel_stats.txt | cut -c1-170
