#!/bin/bash
# one-rank numbers of the one-sample-per-rank FNO3d step after the native sharded axis passes
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s10; mkdir -p $O
cd $GRAFT_REPO_ROOT
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --stage-iters 2"
python bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 $Q > $O/ms_b1.json 2> $O/ms_b1.err
python bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --emulate-world 8 $Q > $O/ms_b1_emu8.json 2> $O/ms_b1_emu8.err
python bench.py --workload fno3d_128_m32_c32_b1 --graph $Q > $O/plain_b1.json 2> $O/plain_b1.err
python bench.py --workload fno3d_128_m32_c32_b8 $Q > $O/plain_b8.json 2> $O/plain_b8.err
python - <<'PY'
import json
for f in ("ms_b1", "ms_b1_emu8", "plain_b1", "plain_b8"):
    try:
        d = json.loads(open("gpurun_out/r05_s10/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], d["config"]["launch"][:50], {k: v["ms"] for k, v in d["stages"].items()})
    except Exception as e:
        print(f, "failed", e); print(open("gpurun_out/r05_s10/%s.err" % f).read()[-800:])
PY
