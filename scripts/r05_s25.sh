#!/bin/bash
# round 5, session 2: k_fft2d_inv_mx with the per-r operand fragments in LDS (coefficients split once, no rotation)
mkdir -p gpurun_out/r05_s25
for h in 256 128 64; do python scripts/mx_ifft_ab.py $h 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_s25/ab.txt; done
cat gpurun_out/r05_s25/ab.txt
for i in 1 2; do
python bench.py --io bf16 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mx  ', d['ms_per_step'], d['value'])"
SC_PLAN_NO_MX_FFT=1 python bench.py --io bf16 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('valu', d['ms_per_step'], d['value'])"
done
