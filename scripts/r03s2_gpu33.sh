#!/bin/bash
O=gpurun_out/s2am; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for wl in fno2d_64_m32_c64_b64 fno3d_64_m16_c32_b8; do
  for v in eager graph; do
    fl=""; [ $v = graph ] && fl="--graph"
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --workload $wl $fl --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/bench_${wl}_$v.json 2> $O/bench_${wl}_$v.err || tail -5 $O/bench_${wl}_$v.err
    python -c "
import json; d=json.load(open('$O/bench_${wl}_$v.json')); print('$wl $v', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], d['config'].get('engine_path'), d['config'].get('launch'))"
  done
done 2>&1 | tee $O/summary.txt
