#!/bin/bash
# round 3, GPU call 20: factor-matrix kernels alone, ablations
O=gpurun_out/r3v; mkdir -p $O
TAG="full" timeout 120 python scripts/fmx_time.py 2>&1 | tail -3 >> $O/fmx_time.txt
for a in 1 2 3; do TAG="abl=$a" SC_TK_ABL=$a timeout 120 python scripts/fmx_time.py 2>&1 | tail -3 >> $O/fmx_time.txt; done
TAG="valu" SC_FMX_OFF=1 timeout 120 python scripts/fmx_time.py 2>&1 | tail -3 >> $O/fmx_time.txt
cat $O/fmx_time.txt
TAG="mx" timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 > $O/tucker_time.txt
TAG="abl3" SC_TK_ABL=3 timeout 120 python scripts/tucker_time.py 2>&1 | tail -2 >> $O/tucker_time.txt; cat $O/tucker_time.txt
