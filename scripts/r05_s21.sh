#!/bin/bash
# k_f2p_col_inv_w1024<true>: the sixteen inputs of a thread requested one block ahead (libsc_engine_nopf.so = the build before)
cd /tmp && export TMPDIR=/tmp
for v in "" _nopf "" _nopf; do
  rm -rf /tmp/pf$v
  SC_ENGINE_LIB=$GRAFT_REPO_ROOT/neuraloperator_amd/libsc_engine$v.so LAYER_SHAPE=4,128,1024,1024,256,256 LAYER_REPS=4 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf$v -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
  echo "== libsc_engine$v.so"; python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py /tmp/pf$v | grep -E "k_f2p_col_inv" | cut -c1-150
done
cd $GRAFT_REPO_ROOT
Q="--steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference"
for v in "" _nopf "" _nopf; do
  SC_ENGINE_LIB=$GRAFT_REPO_ROOT/neuraloperator_amd/libsc_engine$v.so python bench.py --workload fno2d_1024_m256_c128_b4 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step libsc_engine$v', d['ms_per_step'], {k: v['ms'] for k, v in d['stages'].items() if 'transform' in k})"
done
python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -q -k "c5 or 1024" 2>&1 | grep -E "passed|failed"
