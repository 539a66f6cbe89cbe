#!/bin/bash
# round 3, GPU call 19: factor-matrix products and mode-summed contractions on the matrix cores (sc_kernels_fmx.h)
O=gpurun_out/r3t; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -x -q -k "tucker or tfno or factor or galore or cp or variants" 2>&1 | tail -3) > $O/pytest_fmx.log
cat $O/pytest_fmx.log
for v in on off on off; do
  echo "fmx: $v" >> $O/tfno_time.txt
  if [ $v = off ]; then export SC_FMX_OFF=1; else unset SC_FMX_OFF; fi
  (timeout 200 python scripts/tfno_time.py factorized 2>&1 | tail -1) >> $O/tfno_time.txt
done
unset SC_FMX_OFF
cat $O/tfno_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/tfno_kernel_stats_fmx.txt 2>&1
head -22 $O/tfno_kernel_stats_fmx.txt | cut -c1-170
TAG="mx" timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 > $O/tucker_time.txt; cat $O/tucker_time.txt
