#!/bin/bash
# round 3, GPU call 22: register-blocked 16x16x4 tiles (one operand shared by 3-4 tiles, tied accumulators)
O=gpurun_out/r3x; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -x -q -k "tucker or tfno or factor or galore or cp or variants" 2>&1 | tail -3) > $O/pytest_fmx.log
cat $O/pytest_fmx.log
TAG="full" timeout 120 python scripts/fmx_time.py 2>&1 | tail -3 >> $O/fmx_time.txt
TAG="abl=1" SC_TK_ABL=1 timeout 120 python scripts/fmx_time.py 2>&1 | tail -3 >> $O/fmx_time.txt
cat $O/fmx_time.txt
TAG="mx" timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 > $O/tucker_time.txt
TAG="abl1" SC_TK_ABL=1 timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/tucker_time.txt; cat $O/tucker_time.txt
(timeout 200 python scripts/tfno_time.py factorized 2>&1 | tail -1) > $O/tfno_time.txt; cat $O/tfno_time.txt
