#!/bin/bash
# round 3, session 2, GPU call 28: k_f2p_c2r with the first item through tracked loads -- two-pass parity cases, step times
O=gpurun_out/s2af; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "f2p or two_pass or 192 or 1024 or 96 or 160 or 384 or 640 or C5 or 64" 2>&1 | grep -E "passed|failed|rror" | tail -3
for wl in fno2d_1024_m256_c128_b4 fno2d_192_m64_c64_b32; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "
import json; d=json.load(open('$O/bench_$wl.json')); print('$wl', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'])"
done
