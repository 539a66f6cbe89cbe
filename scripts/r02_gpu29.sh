#!/bin/bash
# round 2, GPU call 29 (round end): full GPU tier, smoke, the driver's bench command, the other BASELINE shapes,
# rocprof kernel stats of the bench command
O=gpurun_out/r2ac; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_full.log 2>&1
grep -E "passed|failed|error" $O/pytest_full.log | tail -3
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tail -9) > $O/smoke.log; tail -2 $O/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2ac/bench_default.json"))
print({k: d[k] for k in ("ms_per_step", "value", "step_roofline")}); print(d["cold_start"]["ms_per_step"], d["clock_settle"]["steps"])
print(d["roofline"]); print(d["stages"]); print(d.get("extra")); print(d.get("gpu_reference_baseline")); print(d.get("cpu_baseline"))
PY
timeout 200 python bench.py --io bf16 --no-cpu-baseline --no-gpu-reference --no-extras > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 200 python bench.py --workload fno3d_128_m32_c32_b8 --no-cpu-baseline --no-gpu-reference > $O/bench_3d.json 2> $O/bench_3d.err
timeout 200 python bench.py --workload fno2d_1024_m256_c128_b4 --no-cpu-baseline --no-gpu-reference > $O/bench_1024.json 2> $O/bench_1024.err
timeout 200 python bench.py --workload fno2d_128_m32_c64_b32 --no-cpu-baseline --no-gpu-reference > $O/bench_128.json 2> $O/bench_128.err
for f in bf16 3d 1024 128; do python -c "
import json,sys
d=json.load(open('gpurun_out/r2ac/bench_$f.json')); print('$f', d['ms_per_step'], d['cold_start']['ms_per_step'], d['step_roofline']['frac_of_8TBs'], d['roofline']['kernel'], d['roofline']['frac'])"; done
(timeout 200 python scripts/tfno_time.py 2>&1 | tail -2) > $O/tfno_time.txt; cat $O/tfno_time.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-gpu-reference > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
head -8 $O/kernel_stats.txt | cut -c1-60,115-170
python -c "
import json; d=json.load(open('gpurun_out/r2ac/bench_prof.json')); print('under rocprof', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['ms_per_launch'], d['stages'])"
