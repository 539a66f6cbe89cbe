#!/bin/bash
# round 3, session 2, GPU call 20: persistent k_f2p_c2r (untracked prefetch) -- two-pass parity cases, 1024^2 transforms, 192^2 / 1024^2 steps
O=gpurun_out/s2t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "f2p or two_pass or 192 or 1024 or 96 or 160 or 384 or 640 or C5" 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python scripts/f2p_time.py 2>&1 | tail -6 | cut -c1-200 | tee $O/f2p_time.txt
for wl in fno2d_1024_m256_c128_b4 fno2d_192_m64_c64_b32; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "
import json; d=json.load(open('$O/bench_$wl.json')); print('$wl', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k:v['ms'] for k,v in d['stages'].items()})"
done
