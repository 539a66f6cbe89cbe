#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --stage-iters 2 --settle-ms 0"
rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -o run -- python $GRAFT_REPO_ROOT/bench.py --workload fno3d_128_m32_c32_b8 $Q > /dev/null 2> $O/prof.err
python $GRAFT_REPO_ROOT/scripts/step_timeline.py /tmp/p1 k_pl128_fwd > $O/b8_timeline.txt 2>&1
cat $O/b8_timeline.txt
