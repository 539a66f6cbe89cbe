#!/bin/bash
# round 2, GPU call 11: first-axis 128-point line kernel (FNO3d), TFNO kernel breakdown
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5) > $O/pytest.log
cat $O/pytest.log
for w in fno3d_128_m32_c32_b8; do
  (timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-extras 2>&1 | tail -1) > $O/bench_$w.json
  python - <<PY
import json
d = json.load(open("$O/bench_$w.json"))
print("$w", d["ms_per_step"], "ms/step", d["step_roofline"]["frac_of_8TBs"], {k: v["ms"] for k, v in d["stages"].items()})
PY
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --workload fno3d_128_m32_c32_b8 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/fno3d_kernel_stats.txt 2>&1
head -12 $O/fno3d_kernel_stats.txt | cut -c1-170
cd /tmp; rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py > $GRAFT_REPO_ROOT/$O/tfno_time.txt 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/tfno_kernel_stats.txt 2>&1
tail -3 $O/tfno_time.txt; head -24 $O/tfno_kernel_stats.txt | cut -c1-200
