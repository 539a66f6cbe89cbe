#!/bin/bash
# round 3, session 2, GPU call 17: one-pass small-batch backward pair -- at-config parity (C5), 1024^2 step time with / without it
O=gpurun_out/s2q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_at_config.py -m gpu -x -q -k "C5 or 1024" 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in pair nopair; do
  if [ $v = nopair ]; then export SC_SB_NO_PAIR=1; else unset SC_SB_NO_PAIR; fi
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --workload fno2d_1024_m256_c128_b4 --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/bench_1024_$v.json 2> $O/bench_1024_$v.err
  python -c "
import json; d=json.load(open('$O/bench_1024_$v.json')); print('$v', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k:v['ms'] for k,v in d['stages'].items()})"
done
