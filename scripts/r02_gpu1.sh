#!/bin/bash
# round 2, GPU call 1: streamed contraction kernel (parity + A-B), at-config parity tests, bench + rocprof
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
P=neuraloperator_amd
(timeout 300 python scripts/gemm8_ab.py $P/libsc_engine.so $P/libsc_engine_d4.so $P/libsc_engine_d8.so $P/libsc_engine_nomfma.so $P/libsc_engine_nostore.so 2>&1 | tail -20) > $O/gemm8_ab.txt
(timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "mfma_contractions or full_size" 2>&1 | tail -5) > $O/pytest_gemm.log
(timeout 600 python -m pytest tests/test_gpu_at_config.py -q -s 2>&1 | tail -25) > $O/pytest_at_config.log
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6) > $O/pytest_parity.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err)
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
cat $O/gemm8_ab.txt $O/pytest_gemm.log $O/pytest_at_config.log $O/pytest_parity.log; head -c 700 $O/bench_default.json; echo; head -14 $O/kernel_stats.txt
