#!/bin/bash
# session 8: the matrix-core row pass of the bf16 forward transform -- A-B of its load paths / ablations, parity
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s8; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/mx_ab4.txt
for v in "" _mxnl _mxnm _mxnc _mxnn; do
  echo "== libsc_engine$v.so" >> $O/mx_ab4.txt
  SC_ENGINE_LIB=$GRAFT_REPO_ROOT/neuraloperator_amd/libsc_engine$v.so python scripts/mx_fft_ab.py 256 2>&1 | grep -v amdgpu.ids >> $O/mx_ab4.txt
done
python scripts/mx_fft_ab.py 128 2>&1 | grep -v amdgpu.ids >> $O/mx_ab4.txt
python scripts/mx_fft_ab.py 64 2>&1 | grep -v amdgpu.ids >> $O/mx_ab4.txt
cat $O/mx_ab4.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "bf16 or sharded_transforms" 2>&1 | tail -3
python -m pytest tests/test_gpu_at_config.py -q -x -k "bf16" 2>&1 | tail -3
