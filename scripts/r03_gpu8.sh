#!/bin/bash
# round 3, GPU call 8: TFNO step with the prefetch ring in k_modegemm_bfac and four steps of loads in flight in
# k_modegemm_msum; bf16-I/O and default bench lines with the depth-2 forward kernel
O=gpurun_out/r3h; mkdir -p $O
timeout 200 python scripts/tfno_time.py factorized > $O/tfno_time.txt 2>&1; tail -1 $O/tfno_time.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tfno -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1)
python scripts/rocprof_summary.py /tmp/prof_tfno > $O/tfno_kernel_stats.txt 2>&1; head -16 $O/tfno_kernel_stats.txt | cut -c1-170
python bench.py --no-cpu-baseline --no-gpu-reference --no-extras > $O/bench_default_noextras.json 2> $O/bench_default_noextras.err
python bench.py --no-cpu-baseline --no-gpu-reference --no-extras --io bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python - <<PY
import json
for f in ("bench_default_noextras", "bench_bf16"):
    d = json.load(open("$O/" + f + ".json"))
    print(f, "ms/step", d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], "step frac", d["step_roofline"]["frac_of_8TBs"], {k: v["ms"] for k, v in d["stages"].items()})
PY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "golden or tucker or tfno or bf16 or full_size" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
