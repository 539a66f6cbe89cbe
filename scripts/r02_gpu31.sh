#!/bin/bash
# round 2, GPU call 31: full GPU tier on the final library (small-batch rule, pair tests), FNO3d bench line
O=gpurun_out/r2ae; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_full.log 2>&1
grep -E "passed|failed|error" $O/pytest_full.log | tail -3
timeout 200 python bench.py --workload fno3d_128_m32_c32_b8 --no-cpu-baseline --no-gpu-reference > $O/bench_3d.json 2> $O/bench_3d.err
python -c "
import json
d=json.load(open('gpurun_out/r2ae/bench_3d.json')); print('3d', d['ms_per_step'], d['cold_start']['ms_per_step'], d['step_roofline']['frac_of_8TBs'], d['roofline']['kernel'], d['roofline']['frac'], d['stages'])"
