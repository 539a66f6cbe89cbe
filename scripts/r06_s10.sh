#!/bin/bash
# round 6, GPU call 10: baseline of the block (fused path only) + GPU tier + default bench line
cd /root/repo; O=gpurun_out/r06_s10; mkdir -p $O; export TMPDIR=/tmp
python scripts/block_time.py > $O/block_time.txt 2>&1
BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py > /dev/null 2>&1
python scripts/rocprof_summary.py $O/prof > $O/block_kernel_stats.txt 2>&1; rm -rf $O/prof
BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py 32 128 256 256 64 > /dev/null 2>&1
python scripts/rocprof_summary.py $O/prof > $O/block128_kernel_stats.txt 2>&1; rm -rf $O/prof
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/block_time.txt; head -20 $O/block_kernel_stats.txt; head -20 $O/block128_kernel_stats.txt; cat $O/pytest.log; tail -c 1600 $O/bench_default.json
