#!/bin/bash
# Round-end GPU pass (one gpurun call): parity tests, smoke, bench lines, rocprof kernel stats, PMC traffic.
# Usage on the GPU box: bash scripts/round_end_gpu.sh   (writes gpurun_out/s3z)
set -x
O=gpurun_out/s3z; mkdir -p $O
(timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12) > $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --io bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --workload fno3d_128_m32_c32_b8 --no-cpu-baseline > $O/bench_3d.json 2> $O/bench_3d.err
python scripts/tfno_time.py > $O/tfno.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
python scripts/rocprof_summary.py $O/prof > $O/kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o run -- python scripts/fft_one.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o run -- python scripts/fft_one.py > /dev/null 2>&1
python scripts/pmc_summary.py $O/pmc_f $O/pmc_w > $O/pmc_f32.txt 2>&1
rm -rf $O/prof $O/pmc_f $O/pmc_w
cat $O/pytest.log $O/smoke.log; head -c 600 $O/bench_default.json; echo; cat $O/kernel_stats.txt | head -12; cat $O/pmc_f32.txt; tail -5 $O/tfno.txt; head -c 400 $O/bench_3d.json
