#!/bin/bash
# session 8: the matrix-core row pass of the bf16 forward transform -- A-B of its load paths / ablations, parity
O=$GRAFT_REPO_ROOT/gpurun_out/r05_s8; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/mx_ab5.txt
for v in "" _mxnp _mxnomix _mxt2; do
  echo "== libsc_engine$v.so" >> $O/mx_ab5.txt
  SC_ENGINE_LIB=$GRAFT_REPO_ROOT/neuraloperator_amd/libsc_engine$v.so python scripts/mx_fft_ab.py 256 2>&1 | grep -v amdgpu.ids >> $O/mx_ab5.txt
done
python scripts/mx_fft_ab.py 128 2>&1 | grep -v amdgpu.ids >> $O/mx_ab5.txt
python scripts/mx_fft_ab.py 64 2>&1 | grep -v amdgpu.ids >> $O/mx_ab5.txt
cat $O/mx_ab5.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "bf16 or sharded_transforms" 2>&1 | tail -3
python -m pytest tests/test_gpu_at_config.py -q -x -k "bf16" 2>&1 | tail -3
