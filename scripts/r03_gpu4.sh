#!/bin/bash
# round 3, GPU call 4: the two-pass factorised route on lines of 32 P points (64 .. 640 per axis) against the
# direct-DFT passes it replaces (--plan-flags 32 = SC_PLAN_NO_F2P_SMALL), explicit LDS read widths in the plane /
# two-pass kernels (FNO3d 128^3, 1024^2), parity of the new widths
O=gpurun_out/r3d; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-gpu-reference --no-extras --steps 10 --warmup 3"
for wl in fno2d_64_m32_c64_b64 fno2d_192_m64_c64_b32; do
  for fl in 0 32; do
    $B --workload $wl --plan-flags $fl > $O/bench_${wl}_flags$fl.json 2> $O/bench_${wl}_flags$fl.err
    python - <<PY
import json
d = json.load(open("$O/bench_${wl}_flags$fl.json"))
print("$wl flags=$fl", d["config"]["engine_path"], "ms/step", d["ms_per_step"], "step frac", d["step_roofline"]["frac_of_8TBs"], {k: v["ms"] for k, v in d["stages"].items()})
PY
  done
done 2>&1 | tee $O/widths_ab.txt
for wl in fno3d_128_m32_c32_b8 fno2d_1024_m256_c128_b4 fno2d_128_m32_c64_b32; do
  $B --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err
  python - <<PY
import json
d = json.load(open("$O/bench_$wl.json"))
print("$wl", d["config"]["engine_path"], "ms/step", d["ms_per_step"], "step frac", d["step_roofline"]["frac_of_8TBs"], {k: v["ms"] for k, v in d["stages"].items()})
PY
done 2>&1 | tee $O/other_workloads.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_vs_oracle or factorised_route or test_golden" > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
timeout 600 python -m pytest tests/test_gpu_at_config.py -m gpu -x -q -k "C4 or C5" > $O/gpu_at_config.txt 2>&1; tail -3 $O/gpu_at_config.txt
