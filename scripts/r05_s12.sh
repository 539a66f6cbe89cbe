#!/bin/bash
# work-item order of the small-batch contraction kernels (SC_SB_ALT_ORDER: bit 0 flips the single launches, bit 1 the pair)
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s12; mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference"
for alt in 0 2 1 0 2; do
  SC_SB_ALT_ORDER=$alt python bench.py --workload fno2d_1024_m256_c128_b4 $Q > $O/b_$alt.json 2> $O/b_$alt.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_s12/b_$alt.json").read().strip().splitlines()[-1])
print("1024^2 alt $alt", d["ms_per_step"], {k: v["ms"] for k, v in d["stages"].items()})
PY
done
for cfg in "0 4" "0 8" "1 8" "2 8" "3 8"; do
  set -- $cfg
  SC_SB_ALT_ORDER=$1 SC_SB_MAX=$2 python bench.py --workload fno3d_128_m32_c32_b8 $Q > $O/c.json 2> $O/c.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_s12/c.json").read().strip().splitlines()[-1])
print("fno3d b8 alt $1 sb_max $2", d["ms_per_step"], {k: v["ms"] for k, v in d["stages"].items() if "contract" in k})
PY
done
