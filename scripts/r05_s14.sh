#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s14; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t && LAYER_KIND=tucker LAYER_REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof_t | cut -c1-150 | head -22
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-gpu-reference 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['ms_per_step']); print('tfno', d['extra']['tfno_rank01']['ms_per_step'])"
python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -q -k "tucker or tfno or cp or factor" 2>&1 | grep -E "passed|failed"
