#!/bin/bash
cd $GRAFT_REPO_ROOT
Q="--steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference"
for v in "" _sbnoa _sbnob _sbnoab ""; do
  SC_ENGINE_LIB=$GRAFT_REPO_ROOT/neuraloperator_amd/libsc_engine$v.so python bench.py --workload fno2d_1024_m256_c128_b4 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('libsc_engine$v', d['ms_per_step'], {k: v['ms'] for k, v in d['stages'].items() if 'contract' in k})"
done
