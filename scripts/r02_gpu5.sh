#!/bin/bash
# round 2, GPU call 5: software-pipelined contraction stage (IL) + two-queue transforms
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
P=neuraloperator_amd
(timeout 300 python scripts/gemm8_ab.py $P/libsc_engine.so $P/libsc_engine_v4.so $P/libsc_engine_v5.so $P/libsc_engine_v3.so 2>&1 | grep "fno2d_256\|B32 C128") > $O/gemm8_ab.txt
(timeout 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6) > $O/pytest_parity.log
for v in 0 32; do
  timeout 200 python bench.py --no-cpu-baseline --no-extras --no-gpu-reference --plan-flags $v > $O/bench_flags$v.json 2> $O/bench_flags$v.err
done
timeout 200 python bench.py --no-cpu-baseline --no-extras --no-gpu-reference --plan-flags 0 > $O/bench_flags0b.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --no-extras --no-gpu-reference --plan-flags 32 > $O/bench_flags32b.json 2>/dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gpu-reference > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err)
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
cat $O/gemm8_ab.txt $O/pytest_parity.log
for f in flags0 flags32 flags0b flags32b; do python -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k: v['ms'] for k, v in d['stages'].items()})"; done
head -12 $O/kernel_stats.txt
