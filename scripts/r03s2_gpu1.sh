#!/bin/bash
# round 3, session 2, GPU call 1: state check on a fresh box -- the GPU tier, smoke, the default bench line
O=gpurun_out/s2a; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/pytest.log; cat $O/pytest.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4) > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
head -c 1500 $O/bench_default.json; echo; tail -3 $O/bench_default.err
