#!/bin/bash
# round 3, session 2, GPU call 7: wall time of the driver's bench command; the new mode-parallel GPU cases; GPU tier summary line
O=gpurun_out/s2g; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; cat $O/bench_time.txt
head -c 400 $O/bench_default.json; echo
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mode_parallel" 2>&1 | grep -E "passed|failed|error" | tail -3
