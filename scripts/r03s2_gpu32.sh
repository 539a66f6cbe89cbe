#!/bin/bash
# round 3, session 2: 64 x 64 plane kernels -- parity (ORACLE_CASES that land on them, route test) and step time
O=gpurun_out/s2ak; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "oracle or route or emu_matches" 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
for wl in fno3d_64_m16_c32_b8 fno2d_64_m32_c64_b64; do
  for v in new old; do
    if [ $v = old ]; then export SC_PLAN_NO_PL64=1; else unset SC_PLAN_NO_PL64; fi
    timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/bench_${wl}_$v.json 2> $O/bench_${wl}_$v.err
    python -c "
import json; d=json.load(open('$O/bench_${wl}_$v.json')); print('$wl $v', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], d['config'].get('engine_path'), {k:v['ms'] for k,v in d['stages'].items()})"
  done
done 2>&1 | tee $O/summary.txt
