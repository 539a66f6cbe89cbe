#!/bin/bash
# round 5, session 2: first measurement of k_fft2d_inv_mx (bf16 output, matrix-core row pass) -- A-B against
# k_fft2d_inv3<256, sc_bf16>, the bf16 step, the bf16 parity tests
mkdir -p gpurun_out/r05_s23
python scripts/mx_ifft_ab.py 256 > gpurun_out/r05_s23/ab256.txt 2>&1
python scripts/mx_ifft_ab.py 128 >> gpurun_out/r05_s23/ab256.txt 2>&1
python scripts/mx_ifft_ab.py 64 >> gpurun_out/r05_s23/ab256.txt 2>&1
cat gpurun_out/r05_s23/ab256.txt
for i in 1 2; do
python bench.py --io bf16 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mx  ', d['ms_per_step'], d['value'])"
SC_PLAN_NO_MX_FFT=1 python bench.py --io bf16 --no-extras --no-pmc --no-cpu-baseline --no-gpu-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('valu', d['ms_per_step'], d['value'])"
done
python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -5
