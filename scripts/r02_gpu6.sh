#!/bin/bash
# round 2, GPU call 6: full GPU test tier, the bench line with all legs, rocprof stats, PMC traffic of every stage
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
(timeout 200 python scripts/gemm_diag.py 2>&1 | tail -12) > $O/gemm_diag.txt
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/pytest_gpu.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -9) > $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gpu-reference > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
python scripts/pmc_summary.py /tmp/pmc_f /tmp/pmc_w > $O/pmc_traffic_raw.txt 2>&1
cat $O/gemm_diag.txt $O/pytest_gpu.log $O/smoke.log; head -c 2500 $O/bench_default.json; echo; head -12 $O/kernel_stats.txt; grep -A3 "k_fft2d\|k_modegemm_dma\|k_bias" $O/pmc_traffic_raw.txt | head -60
