#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s13; mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference"
for sh in 0 1 2 3 4 0; do
  SC_SB_SHAPE=$sh python bench.py --workload fno2d_1024_m256_c128_b4 $Q > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_s13/b.json").read().strip().splitlines()[-1])
print("shape $sh", d["ms_per_step"], {k: v["ms"] for k, v in d["stages"].items() if "contract" in k})
PY
done
