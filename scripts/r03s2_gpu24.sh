#!/bin/bash
# round 3, session 2, GPU call 24: plane kernels with the twiddle table in LDS -- FNO3d parity + step time
O=gpurun_out/s2x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -m gpu -x -q -k "C4 or fno3d or plane or 128" 2>&1 | grep -E "passed|failed|rror" | tail -3
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --workload fno3d_128_m32_c32_b8 --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/bench_3d.json 2> $O/bench_3d.err
python -c "
import json; d=json.load(open('$O/bench_3d.json')); print(d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k:v['ms'] for k,v in d['stages'].items()})"
done
