#!/bin/bash
# round 2, GPU call 9: two-pass factorised transforms for 1024^2 -- agreement with the direct-DFT route, timing,
# chunk-size variants, kernel stats, at-config parity, large-grid bench line
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python scripts/f2p_time.py neuraloperator_amd/libsc_engine.so neuraloperator_amd/libsc_engine_c32.so neuraloperator_amd/libsc_engine_c192.so neuraloperator_amd/libsc_engine_c4096.so 2>&1 | tail -14) > $O/f2p_time.txt
cat $O/f2p_time.txt
(timeout 600 python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5) > $O/pytest.log
cat $O/pytest.log
(timeout 300 python bench.py --workload fno2d_1024_m256_c128_b4 --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-reference --no-extras 2>&1 | tail -1) > $O/bench_1024.json
cat $O/bench_1024.json
cd /tmp
F2P_IMAGES=512 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/f2p_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/f2p_kernel_stats.txt 2>&1
head -14 $O/f2p_kernel_stats.txt
