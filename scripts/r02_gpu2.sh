#!/bin/bash
# round 2, GPU call 2: 128-byte-segment streamed contraction (A-B vs 64-byte form and generation 1), parity, bench
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
P=neuraloperator_amd
(timeout 300 python scripts/gemm8_ab.py $P/libsc_engine.so $P/libsc_engine_f64.so $P/libsc_engine_d3.so $P/libsc_engine_nomfma.so $P/libsc_engine_nostore.so 2>&1 | tail -20) > $O/gemm8_ab.txt
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6) > $O/pytest_parity.log
(timeout 300 python -m pytest tests/test_gpu_at_config.py -x -q -k "C2 or C5" 2>&1 | tail -6) > $O/pytest_at_config.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err)
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
cat $O/gemm8_ab.txt $O/pytest_parity.log $O/pytest_at_config.log; head -c 400 $O/bench_default.json; echo; head -12 $O/kernel_stats.txt
