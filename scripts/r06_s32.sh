#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s32; mkdir -p $O; export TMPDIR=/tmp
BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py 4 128 1024 1024 256 > /dev/null 2>&1
python scripts/rocprof_summary.py $O/prof > $O/block_c4_kernel_stats.txt 2>&1; rm -rf $O/prof
BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py 8 128 256 256 64 > /dev/null 2>&1
python scripts/rocprof_summary.py $O/prof > $O/block_128_kernel_stats.txt 2>&1; rm -rf $O/prof
head -24 $O/block_c4_kernel_stats.txt; head -22 $O/block_128_kernel_stats.txt
