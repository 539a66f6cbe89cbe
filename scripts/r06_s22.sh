#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s22; mkdir -p $O; export TMPDIR=/tmp
{ python scripts/block_time.py; python scripts/block_time.py; } 2>&1 | grep -v amdgpu.ids > $O/block_time.txt
BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py > /dev/null 2>&1
python scripts/rocprof_summary.py $O/prof > $O/block_kernel_stats.txt 2>&1; rm -rf $O/prof
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block_pass.py -m gpu -x -q -k "block or pointwise or two_pass or hidden" 2>&1 | tail -3) >> $O/block_time.txt
cat $O/block_time.txt; head -14 $O/block_kernel_stats.txt
