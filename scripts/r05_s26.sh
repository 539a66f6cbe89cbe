#!/bin/bash
# round 5, session 2: k_fft2d_inv_mx as shipped (operand fragments in LDS): A-B at the three heights it serves / does not
# serve, ablation (measurement builds), the bf16 step against SC_PLAN_NO_MX_FFT on the same box, rocprof of the bf16 step
mkdir -p gpurun_out/r05_s26
O=gpurun_out/r05_s26/ab.txt
for h in 256 128; do python scripts/mx_ifft_ab.py $h 2>&1 | grep -v amdgpu.ids >> $O; done
for t in _mxi_nocol _mxi_nostore _mxi_norow _mxi_nomfma; do
  echo "== measurement build ${t}" >> $O
  SC_ENGINE_LIB=$PWD/neuraloperator_amd/libsc_engine$t.so python scripts/mx_ifft_ab.py 256 2>&1 | grep "^mx" >> $O
done
for i in 1 2 3; do
python bench.py --io bf16 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 step, both transforms on the matrix cores:', d['ms_per_step'], 'ms', d['value'], 'samples/s')" >> $O
SC_PLAN_NO_MX_FFT=1 python bench.py --io bf16 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 step, SC_PLAN_NO_MX_FFT (vector-ALU kernels):  ', d['ms_per_step'], 'ms', d['value'], 'samples/s')" >> $O
done
cat $O
