#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s15; mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference"
for m in "" "--graph" "" "--graph"; do
  python bench.py $m $Q > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_s15/b.json").read().strip().splitlines()[-1])
print("mode '$m'", d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], d["config"]["launch"][:60])
PY
done
for m in "" "--graph"; do
  python bench.py --io bf16 $m $Q > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_s15/b.json").read().strip().splitlines()[-1])
print("bf16 mode '$m'", d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], d["config"]["launch"][:60])
PY
done
