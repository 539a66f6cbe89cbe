#!/bin/bash
O=gpurun_out/s2n; mkdir -p $O
for pass in 1 2; do for b in base st4 st6 qt2st4 qt2st8 qt8st2 qt1st8; do timeout 120 scripts/sb_$b.bin; done; done > $O/sb.txt 2>&1; cat $O/sb.txt
