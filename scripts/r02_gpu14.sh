#!/bin/bash
# round 2, GPU call 14: XCD-aware panel blocks in the column kernels (A/B), waves-over-modes VALU contraction (A/B)
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python scripts/f2p_time.py neuraloperator_amd/libsc_engine_noxcd.so neuraloperator_amd/libsc_engine.so neuraloperator_amd/libsc_engine_noxcd.so neuraloperator_amd/libsc_engine.so 2>&1 | tail -9) > $O/f2p_time.txt
cat $O/f2p_time.txt
cp neuraloperator_amd/libsc_engine.so /tmp/prod.so
for lib in libsc_engine_nowm.so prod libsc_engine_nowm.so prod; do
  if [ $lib = prod ]; then cp /tmp/prod.so neuraloperator_amd/libsc_engine.so; else cp neuraloperator_amd/$lib neuraloperator_amd/libsc_engine.so; fi
  (timeout 300 python bench.py --workload fno2d_1024_m256_c128_b4 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-extras 2>&1 | tail -1) > $O/bench_1024_$lib.json
  python -c "
import json; d=json.load(open('$O/bench_1024_$lib.json')); print('$lib', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k: v['ms'] for k, v in d['stages'].items()})"
done
cp /tmp/prod.so neuraloperator_amd/libsc_engine.so
(timeout 300 python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3) > $O/pytest.log
cat $O/pytest.log
