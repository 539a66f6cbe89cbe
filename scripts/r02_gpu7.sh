#!/bin/bash
# round 2, GPU call 7: epilogue tests + timing, final bench lines (all workloads), rocprof stats, PMC traffic
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5) > $O/pytest_parity.log
(timeout 200 python scripts/fourier_layer_time.py 2>&1 | tail -4) > $O/fourier_layer_time.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload fno2d_1024_m256_c128_b4 --no-cpu-baseline --no-extras --no-gpu-reference --steps 5 --warmup 2 > $O/bench_1024.json 2> $O/bench_1024.err
timeout 300 python bench.py --workload fno3d_128_m32_c32_b8 --no-cpu-baseline --no-extras --no-gpu-reference > $O/bench_3d.json 2> $O/bench_3d.err
timeout 300 python bench.py --io bf16 --no-cpu-baseline --no-extras --no-gpu-reference > $O/bench_bf16.json 2> $O/bench_bf16.err
(timeout 200 python scripts/tfno_time.py 2>&1 | tail -6) > $O/tfno.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gpu-reference > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
python scripts/pmc_summary.py /tmp/pmc_f /tmp/pmc_w > $O/pmc_traffic_raw.txt 2>&1
(SC_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 3 --warmup 1 --no-extras > $O/bench_share2.json 2> $O/bench_share2.err; echo "share2 rc=$?" >> $O/bench_share2.err)
cat $O/pytest_parity.log $O/fourier_layer_time.txt $O/tfno.txt
for f in default 1024 3d bf16; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d['value'], d['step_roofline']['frac_of_8TBs'], d['roofline']['kernel'], d['roofline']['frac'], {k: v['ms'] for k, v in d['stages'].items()}); print(d.get('cpu_baseline')); print(d.get('gpu_reference_baseline')); print(d.get('extra'))"; done
head -11 $O/kernel_stats.txt; grep -A2 "k_fft2d\|k_modegemm_dma" $O/pmc_traffic_raw.txt | head -40; tail -2 $O/bench_share2.err; python -c "import json; d=json.load(open('$O/bench_share2.json')); print(d['config']['parallelism'], d['ms_per_step'], d.get('collectives'))"
