#!/bin/bash
# round 3, GPU call 13: TFNO step: reduction-index split of k_modegemm_msum (SC_MSUM_RSPLIT) and workgroup count of the
# mode-factor kernels (SC_TK_WGS), interleaved in one process sequence on one box
O=gpurun_out/r3l; mkdir -p $O
for rep in 1 2; do
for cfg in "1 512" "2 512" "4 512" "8 512" "1 768" "1 1296" "4 1296"; do
  set -- $cfg
  echo -n "SC_MSUM_RSPLIT=$1 SC_TK_WGS=$2: "; SC_MSUM_RSPLIT=$1 SC_TK_WGS=$2 timeout 100 python scripts/tfno_time.py factorized 2>&1 | tail -1
done; done > $O/tfno_tuning.txt 2>&1
cat $O/tfno_tuning.txt
