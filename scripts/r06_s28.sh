#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s28; mkdir -p $O; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5) > $O/gpu_tier.txt; cat $O/gpu_tier.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/gpu_tier.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1600 $O/bench_default.json
