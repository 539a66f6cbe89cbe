#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s27; mkdir -p $O; export TMPDIR=/tmp
{ python scripts/block_time.py; python scripts/block_time.py 8 128 256 256 64; python scripts/block_time.py 4 128 1024 1024 256; } 2>&1 | grep -v amdgpu.ids > $O/block_time.txt
cat $O/block_time.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $O/gpu_tier.txt; cat $O/gpu_tier.txt
