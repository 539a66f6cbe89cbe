#!/bin/bash
# round 2, GPU call 4: small-workgroup shapes of the streamed contraction (several independent workgroups per CU)
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
P=neuraloperator_amd
(timeout 300 python scripts/gemm8_ab.py $P/libsc_engine.so $P/libsc_engine_v3.so $P/libsc_engine_v1.so $P/libsc_engine_v2.so 2>&1 | tail -20) > $O/gemm8_ab.txt
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6) > $O/pytest_parity.log
(timeout 300 python -m pytest tests/test_gpu_at_config.py -x -q -k "C2 or C5 or tfno" 2>&1 | tail -6) > $O/pytest_at_config.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gpu-reference > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err)
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
(SC_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_share2.json 2> $O/bench_share2.err; echo "share2 rc=$?" >> $O/bench_share2.err)
cat $O/gemm8_ab.txt $O/pytest_parity.log $O/pytest_at_config.log; head -c 1500 $O/bench_default.json; echo; head -12 $O/kernel_stats.txt; tail -3 $O/bench_share2.err; head -c 600 $O/bench_share2.json
