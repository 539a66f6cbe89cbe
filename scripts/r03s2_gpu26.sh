#!/bin/bash
# round 3, session 2, GPU call 26: factor-product / factor-gradient workgroup counts rounded to multiples of the unit count
O=gpurun_out/s2aa; mkdir -p $O
for pass in 1 2 3; do
  TAG="multiple-of-units" timeout 120 python scripts/fmx_time.py 2>&1 | tail -3
  TAG="exact (old)" SC_FMX_WGS_EXACT=1 timeout 120 python scripts/fmx_time.py 2>&1 | tail -3
done | tee $O/fmx_wgs.txt
for pass in 1 2 3; do
  timeout 200 python scripts/tfno_time.py factorized 2>&1 | tail -1
  SC_FMX_WGS_EXACT=1 timeout 200 python scripts/tfno_time.py factorized 2>&1 | tail -1 | sed 's/$/  (exact)/'
done | tee $O/tfno_wgs.txt
timeout 600 python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -m gpu -x -q -k "tucker or tfno or cp or galore" 2>&1 | grep -E "passed|failed|rror" | tail -3
