#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s18; mkdir -p $O
{ cd scripts/session/pmlp; ./pmlp_bench nw4.hsaco 4 0; ./pmlp_bench nw4_gelu.hsaco 4 0; ./pmlp_bench nw4.hsaco 4 0; ./pmlp_bench nw4_gelu.hsaco 4 0; cd /root/repo
  python scripts/block_time.py; python scripts/block_time.py 8 128 256 256 64; python scripts/block_time.py 4 128 1024 1024 256; } 2>&1 | grep -v amdgpu.ids > $O/gelu_both.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block_pass.py -m gpu -x -q -k "block or pointwise or two_pass or hidden" 2>&1 | tail -3) >> $O/gelu_both.txt
cat $O/gelu_both.txt
