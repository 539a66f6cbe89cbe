#!/bin/bash
# round 3, GPU call 10: k_modegemm_bfac with the factor matrix in LDS: the TFNO step
O=gpurun_out/r3j; mkdir -p $O
timeout 200 python scripts/tfno_time.py factorized > $O/tfno_time.txt 2>&1; tail -1 $O/tfno_time.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tfno -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1)
python scripts/rocprof_summary.py /tmp/prof_tfno > $O/tfno_kernel_stats.txt 2>&1; head -14 $O/tfno_kernel_stats.txt | cut -c1-170
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "tucker or tfno or golden or spherical" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
