#!/bin/bash
# k_f2p_r2c_w1024 (one wave per row pair, LDS-DMA rows) against k_f2p_r2c<32, 4> (SC_F2P_NO_R2C_W1024=1)
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
  rm -rf /tmp/pf$v
  if [ $v = 1 ]; then export SC_F2P_NO_R2C_W1024=1; else unset SC_F2P_NO_R2C_W1024; fi
  LAYER_SHAPE=4,128,1024,1024,256,256 LAYER_REPS=4 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf$v -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1
  echo "== SC_F2P_NO_R2C_W1024=$v"; python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py /tmp/pf$v | grep -E "k_f2p_r2c" | cut -c1-150
done
cd $GRAFT_REPO_ROOT
Q="--steps 10 --warmup 3 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference"
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export SC_F2P_NO_R2C_W1024=1; else unset SC_F2P_NO_R2C_W1024; fi
  python bench.py --workload fno2d_1024_m256_c128_b4 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step NO_R2C_W1024=$v', d['ms_per_step'], {k: v['ms'] for k, v in d['stages'].items() if 'transform' in k})"
done
unset SC_F2P_NO_R2C_W1024
python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py tests/test_gpu_route_fuzz.py -q -k "c5 or 1024" 2>&1 | grep -E "passed|failed"
