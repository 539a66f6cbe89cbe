#!/bin/bash
# round 3, GPU call 17: Tucker mode-factor kernels alone -- workgroup count sweep
O=gpurun_out/r3q; mkdir -p $O
for w in 256 512 648 768 1296; do
  TAG="wgs=$w" SC_TK_WGS=$w timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/tucker_time.txt
done
TAG="valu" SC_TK_VALU=1 timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/tucker_time.txt
cat $O/tucker_time.txt
for a in 1 2 3; do
  TAG="abl=$a" SC_TK_ABL=$a timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/tucker_abl.txt
done
cat $O/tucker_abl.txt
