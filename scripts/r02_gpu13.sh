#!/bin/bash
# round 2, GPU call 13: round loops with prefetch in the two-pass row kernels (A/B), PMC traffic of the new routes
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python scripts/f2p_time.py neuraloperator_amd/libsc_engine_it1.so neuraloperator_amd/libsc_engine.so neuraloperator_amd/libsc_engine_it4.so neuraloperator_amd/libsc_engine_it1.so neuraloperator_amd/libsc_engine.so 2>&1 | tail -10) > $O/f2p_time.txt
cat $O/f2p_time.txt
(timeout 300 python -m pytest tests/test_gpu_at_config.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3) > $O/pytest.log
cat $O/pytest.log
(timeout 300 python bench.py --workload fno2d_1024_m256_c128_b4 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-extras 2>&1 | tail -1) > $O/bench_1024.json
python -c "
import json; d=json.load(open('$O/bench_1024.json')); print(d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k: v['ms'] for k, v in d['stages'].items()})"
cd /tmp
for wl in "4,128,1024,1024,256,256" "8,32,128,128,128,32,32,32"; do
  tag=$(echo $wl | cut -d, -f3-5 | tr , x)
  LAYER_SHAPE=$wl LAYER_REPS=2 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f_$tag -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
  LAYER_SHAPE=$wl LAYER_REPS=2 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w_$tag -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
  (cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py /tmp/pmc_f_$tag /tmp/pmc_w_$tag > $O/pmc_traffic_raw_$tag.txt 2>&1; head -30 $O/pmc_traffic_raw_$tag.txt | cut -c1-200)
done
