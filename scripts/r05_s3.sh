#!/bin/bash
# round 5, GPU session 3: kernel timeline of the one-sample-per-rank steps (graph replays)
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s3; mkdir -p $O
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --settle-ms 0 --stage-iters 2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -o run -- python $GRAFT_REPO_ROOT/bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 $Q > $O/ms_b1.json 2> $O/ms_b1.err
python $GRAFT_REPO_ROOT/scripts/step_timeline.py /tmp/p1 > $O/ms_b1_timeline.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o run -- python $GRAFT_REPO_ROOT/bench.py --workload fno3d_128_m32_c32_b1 --graph $Q > $O/plain_b1.json 2> $O/plain_b1.err
python $GRAFT_REPO_ROOT/scripts/step_timeline.py /tmp/p2 > $O/plain_b1_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
python -c "
import json
for f in ('ms_b1','plain_b1'):
    try:
        d=json.loads(open('gpurun_out/r05_s3/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['config']['launch'])
    except Exception as e: print(f, 'failed', e)
"
cat $O/ms_b1_timeline.txt; echo; cat $O/plain_b1_timeline.txt
