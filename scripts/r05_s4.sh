#!/bin/bash
O=gpurun_out/r05_s4; mkdir -p $O
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --stage-iters 2"
for a in auto peer; do
  python bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --a2a $a $Q > $O/ms_b1_$a.json 2> $O/ms_b1_$a.err
done
SC_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --workload darcy_16_m12_c32_b4 --a2a peer $Q --settle-ms 0 > $O/share2_peer.json 2> $O/share2_peer.err
python - <<'PY'
import json
for f in ("ms_b1_auto", "ms_b1_peer", "share2_peer"):
    try:
        d = json.loads(open("gpurun_out/r05_s4/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], d["config"]["launch"], "|", d["collectives"]["issued_by"][:60])
    except Exception as e:
        print(f, "failed", e); print(open("gpurun_out/r05_s4/%s.err" % f).read()[-1500:])
PY
python -m pytest tests -m gpu -q --durations=40 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
grep -A45 "slowest" $O/gpu_tests.log | head -60; tail -3 $O/gpu_tests.log
