#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s35; mkdir -p $O; export TMPDIR=/tmp
for s in "32 64 256 256 64" "8 128 256 256 64" "4 128 1024 1024 256"; do
  n=$(echo $s | tr ' ' '_')
  BLOCK_ONLY_FUSED=1 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python scripts/block_time.py $s > /dev/null 2>&1
  { echo "== block_time.py $s (fused path only), rocprofv3 --kernel-trace --stats"; python scripts/rocprof_summary.py $O/prof | head -24; } >> $O/block_kernel_stats_head.txt 2>&1; rm -rf $O/prof
done
cut -c1-160 $O/block_kernel_stats_head.txt
