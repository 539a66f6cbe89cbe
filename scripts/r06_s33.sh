#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s33; mkdir -p $O; export TMPDIR=/tmp
{ python scripts/block_time.py 4 128 1024 1024 256; python scripts/block_time.py 4 128 1024 1024 256; } 2>&1 | grep -v amdgpu.ids > $O/block_time.txt
cat $O/block_time.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block_pass.py tests/test_gpu_at_config.py -m gpu -x -q -k "block or pointwise or two_pass or hidden or c5 or C5 or epilogue" 2>&1 | grep -E "passed|failed" | tail -3)
python bench.py --workload fno2d_1024_m256_c128_b4 --no-extras --no-cpu-baseline --no-gpu-reference --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs[4] step', d['ms_per_step'])"
