#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s34; mkdir -p $O; export TMPDIR=/tmp
{ python scripts/block_time.py; python scripts/block_time.py 8 128 256 256 64; python scripts/block_time.py 4 128 1024 1024 256; } 2>&1 | grep -v amdgpu.ids > $O/block_time.txt
cat $O/block_time.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block_pass.py -m gpu -x -q -k "block or pointwise or two_pass or hidden" 2>&1 | grep -E "passed|failed" | tail -3)
