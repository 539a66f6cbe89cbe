#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s7; mkdir -p $O
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --stage-iters 2"
cd $GRAFT_REPO_ROOT
for a in auto peer; do
  python bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --emulate-world 8 --a2a $a $Q > $O/emu8_$a.json 2> $O/emu8_$a.err
done
python bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --emulate-world 8 --no-graph $Q > $O/emu8_eager.json 2> $O/emu8_eager.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -o run -- python $GRAFT_REPO_ROOT/bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --emulate-world 8 $Q --settle-ms 0 > /dev/null 2> $O/prof.err
python $GRAFT_REPO_ROOT/scripts/step_timeline.py /tmp/p1 > $O/emu8_timeline.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
for f in ("emu8_auto", "emu8_peer", "emu8_eager"):
    try:
        d = json.loads(open("gpurun_out/r05_s7/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], d["config"]["launch"][:60], "|", d["collectives"]["issued_by"][:40])
    except Exception as e:
        print(f, "failed", e); print(open("gpurun_out/r05_s7/%s.err" % f).read()[-1500:])
PY
cat $O/emu8_timeline.txt
python -m pytest tests/test_gpu_bench_launch.py -q 2>&1 | tail -3
