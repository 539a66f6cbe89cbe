#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s6; mkdir -p $O
Q="--steps 20 --warmup 5 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference --stage-iters 2"
cd $GRAFT_REPO_ROOT
for a in auto peer; do
  python bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --a2a $a $Q > $O/ms_b1_$a.json 2> $O/ms_b1_$a.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pp -o run -- python $GRAFT_REPO_ROOT/bench.py --parallel modeshard --workload fno3d_128_m32_c32_b1 --a2a peer $Q --settle-ms 0 > /dev/null 2> $O/prof.err
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/pp > $O/ms_b1_peer_kernel_stats.txt 2>&1
python - <<'PY'
import json
for f in ("ms_b1_auto", "ms_b1_peer"):
    try:
        d = json.loads(open("gpurun_out/r05_s6/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], d["config"]["launch"][:70], "|", d["collectives"]["issued_by"][:50])
    except Exception as e:
        print(f, "failed", e); print(open("gpurun_out/r05_s6/%s.err" % f).read()[-1500:])
PY
head -16 $O/ms_b1_peer_kernel_stats.txt
python -m pytest tests/test_gpu_graph.py tests/test_gpu_bench_launch.py tests/test_gpu_peer_exchange.py "tests/test_gpu_parity.py::test_backward_pair_small_batch_one_pass" -q 2>&1 | tail -3
