#!/bin/bash
# round 3, session 2: 64 x 64 plane kernels -- device test at transform level; planes per workgroup of the forward kernel
O=gpurun_out/s2ao; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plane64.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for pl in 16384 4096; do
  for ppw in 1 2 4 8 1 2 4 8; do
    PL_PLANES=$pl PL_PPW=$ppw ./scripts/pl64_base.bin 50
  done
done 2>&1 | tee $O/ubench_pl64.txt
for ppw in 1 2 4; do
  SC_P64_PPW=$ppw timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --workload fno3d_64_m16_c32_b8 --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('fno3d_64 ppw=$ppw', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
done 2>&1 | tee -a $O/ubench_pl64.txt
