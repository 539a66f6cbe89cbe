#!/bin/bash
# round 3, GPU call 18: Tucker mode-factor kernels, zero-padded LDS / minimal k loop / one-launch reduction
O=gpurun_out/r3r; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -x -q -k "tucker or tfno or factor" 2>&1 | tail -3) > $O/pytest_tucker.log
cat $O/pytest_tucker.log
for w in 512 648; do
  TAG="wgs=$w" SC_TK_WGS=$w timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/tucker_time.txt
done
for a in 1 3; do
  TAG="abl=$a" SC_TK_ABL=$a timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/tucker_time.txt
done
cat $O/tucker_time.txt
(timeout 200 python scripts/tfno_time.py factorized 2>&1 | tail -1) > $O/tfno_time.txt; cat $O/tfno_time.txt
