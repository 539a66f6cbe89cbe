#!/bin/bash
# round 3, session 2: FNO3d 128^3 step with staged (shipped) against direct row loads in k_pl128_fwd, same box, interleaved
O=gpurun_out/s2az; mkdir -p $O
P=neuraloperator_amd
cp $P/libsc_engine.so /tmp/staged.so; cp $P/libsc_engine_direct.so /tmp/direct.so
for i in 1 2 3; do
  for v in staged direct; do
    cp /tmp/$v.so $P/libsc_engine.so
    python bench.py --gpus 1 --steps 20 --warmup 5 --workload fno3d_128_m32_c32_b8 --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > /tmp/b.json 2>/tmp/b.err
    python -c "
import json; d=json.load(open('/tmp/b.json')); print('$v', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k:v['ms'] for k,v in d['stages'].items()})"
  done
done 2>&1 | tee $O/fno3d_staged_ab.txt
cp /tmp/staged.so $P/libsc_engine.so
