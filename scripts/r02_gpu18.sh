#!/bin/bash
# round 2, GPU call 18: the backward pass's two contractions + bias gradient as one launch (k_modegemm_dma_bwd):
# A-B against the launch sequence and two tiles-per-workgroup variants, full GPU tier, default bench line, kernel stats
O=gpurun_out/r2r; mkdir -p $O
export TMPDIR=/tmp
P=neuraloperator_amd
(timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_nopair.so $P/libsc_engine_bpw1.so $P/libsc_engine_bpw2.so 2>&1 | tail -12) > $O/pair_ab.txt
cat $O/pair_ab.txt
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest.log
cat $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
head -c 700 $O/bench_default.json; echo
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gpu-reference > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
head -12 $O/kernel_stats.txt | cut -c1-170
# HBM traffic of every kernel of the step (separate passes per counter, kernel trace only)
cd /tmp
LAYER_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
LAYER_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py /tmp/pmc_f /tmp/pmc_w > $O/pmc_step.txt 2>&1
cat $O/pmc_step.txt | cut -c1-170 | head -12
