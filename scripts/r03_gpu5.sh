#!/bin/bash
# round 3, GPU call 5: the two-pass route with G = 32 / P lines per half-wave against the direct-DFT passes
# (--plan-flags 32) on 64^2, 96^2-like, 192^2 grids; 512 / 1024-point lines (P = 16 now two pairs per half-wave);
# parity of the rewritten kernels
O=gpurun_out/r3e; mkdir -p $O
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
for wl in fno2d_1024_m256_c128_b4; do
  $B --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err
  python - <<PY
import json
d = json.load(open("$O/bench_$wl.json"))
print("$wl", d["config"]["engine_path"], "ms/step", d["ms_per_step"], "step frac", d["step_roofline"]["frac_of_8TBs"], {k: v["ms"] for k, v in d["stages"].items()})
PY
done 2>&1 | tee $O/other_workloads.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_vs_oracle or factorised_route" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 600 python -m pytest tests/test_gpu_at_config.py -m gpu -x -q -k "C5" > $O/gpu_at_config.txt 2>&1; tail -3 $O/gpu_at_config.txt
