#!/bin/bash
# round 3, session 2, GPU call 6: default bench line with the persistent inverse 2-D kernel + GPU parity subset
O=gpurun_out/s2f; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -m gpu -x -q -k "not tucker and not cp and not tt" 2>&1 | tail -4) > $O/pytest.log; cat $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s2f/bench_default.json'))
print(d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], d['cold_start']['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['ms_per_launch'])
for k,v in d['stages'].items(): print(k, v['ms'], v.get('ms_back_to_back'))
for k,v in d['extra'].items(): print(k, {kk:v[kk] for kk in v if kk in ('ms_per_step','frac_of_8TBs','fused_ms','reference_op_sequence_ms')})
PY
tail -3 $O/bench_default.err
