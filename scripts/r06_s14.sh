#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s14; mkdir -p $O; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hidden_128 or two_pass" 2>&1 | tail -3) > $O/pytest.log
python scripts/block_time.py 8 128 256 256 64 > $O/block_time_c128_256.txt 2>&1
python scripts/block_time.py 4 128 1024 1024 256 > $O/block_time_c128_1024.txt 2>&1
BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py 8 128 256 256 64 > /dev/null 2>&1
python scripts/rocprof_summary.py $O/prof > $O/block128_kernel_stats.txt 2>&1; rm -rf $O/prof
BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py 4 128 1024 1024 256 > /dev/null 2>&1
python scripts/rocprof_summary.py $O/prof > $O/block128_1024_kernel_stats.txt 2>&1; rm -rf $O/prof
cat $O/pytest.log $O/block_time_c128_256.txt $O/block_time_c128_1024.txt; head -12 $O/block128_kernel_stats.txt; head -14 $O/block128_1024_kernel_stats.txt
