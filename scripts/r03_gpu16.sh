#!/bin/bash
# round 3, GPU call 16: Tucker mode-factor kernels on the matrix cores (16x16x4 tiles, three real products)
O=gpurun_out/r3p; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -x -q -k "tucker or tfno or factor" 2>&1 | tail -4) > $O/pytest_tucker.log
cat $O/pytest_tucker.log
for v in mx valu mx valu; do
  echo "tucker modes: $v" >> $O/tfno_time.txt
  if [ $v = valu ]; then export SC_TK_VALU=1; else unset SC_TK_VALU; fi
  (timeout 200 python scripts/tfno_time.py factorized 2>&1 | tail -1) >> $O/tfno_time.txt
done
unset SC_TK_VALU
cat $O/tfno_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/tfno_kernel_stats_mx.txt 2>&1
head -22 $O/tfno_kernel_stats_mx.txt | cut -c1-170
