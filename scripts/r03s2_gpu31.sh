#!/bin/bash
# round 3, session 2: step time of the small-grid workloads that are not BASELINE configs
O=gpurun_out/s2aj; mkdir -p $O
for wl in fno3d_64_m16_c32_b8 fno2d_64_m32_c64_b64 fno2d_128_m32_c64_b32; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "
import json; d=json.load(open('$O/bench_$wl.json')); print('$wl', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], d['config'].get('engine_path'), {k:v['ms'] for k,v in d['stages'].items()})"
done
