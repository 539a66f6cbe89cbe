#!/bin/bash
# round 2, GPU call 8: block epilogue with the cheap erf and pre-issued skip loads: tests + timing
O=gpurun_out/r2h; mkdir -p $O
export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "fourier or golden or full_size" 2>&1 | grep -E "passed|failed|Error" | tail -5) > $O/pytest_parity.log
(timeout 200 python scripts/fourier_layer_time.py 2>&1 | tail -4) > $O/fourier_layer_time.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/fourier_layer_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/fourier_layer_kernel_stats.txt 2>&1
cat $O/pytest_parity.log $O/fourier_layer_time.txt; head -16 $O/fourier_layer_kernel_stats.txt
