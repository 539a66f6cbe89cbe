#!/bin/bash
# session 2 (round 5): the bf16 step with the matrix-core forward transform, against the vector-ALU one
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s9; mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference"
python bench.py --io bf16 $Q > $O/bf16_mx.json 2> $O/bf16_mx.err
SC_PLAN_NO_MX_FFT=1 python bench.py --io bf16 $Q > $O/bf16_valu.json 2> $O/bf16_valu.err
python - <<'PY'
import json
for f in ("bf16_mx", "bf16_valu"):
    try:
        d = json.loads(open("gpurun_out/r05_s9/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], d["step_roofline"]["frac_of_8TBs"], {k: v["ms"] for k, v in d["stages"].items()})
    except Exception as e:
        print(f, "failed", e); print(open("gpurun_out/r05_s9/%s.err" % f).read()[-1500:])
PY
