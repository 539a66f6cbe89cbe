#!/bin/bash
# round 5, session 2: ablation of k_fft2d_inv_mx (measurement builds: no column phase / no stores / no row pass / no MFMA)
mkdir -p gpurun_out/r05_s24
for t in "" _mxi_nocol _mxi_nostore _mxi_norow _mxi_nomfma; do
  echo "== variant ${t:-product}" >> gpurun_out/r05_s24/abl.txt
  SC_ENGINE_LIB=$PWD/neuraloperator_amd/libsc_engine$t.so python scripts/mx_ifft_ab.py 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_s24/abl.txt
done
cat gpurun_out/r05_s24/abl.txt
