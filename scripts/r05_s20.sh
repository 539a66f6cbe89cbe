#!/bin/bash
# one sample per rank, 8-rank contraction load, exchanges chunked over channels INSIDE the graph step
cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --stage-iters 2"
for c in "" "--comm-chunks 2 --chunk-dim channels" "--comm-chunks 4 --chunk-dim channels" ""; do
  python bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --emulate-world 8 $c $Q 2>/tmp/e.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks [$c]', d['ms_per_step'], 'cold', d['cold_start']['ms_per_step'], d['config']['launch'][:50], d['collectives'].get('all_to_all_calls_per_step'))
except Exception as e:
    print('chunks [$c] failed', e); print(open('/tmp/e.err').read()[-600:])"
done
