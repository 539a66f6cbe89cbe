#!/bin/bash
O=gpurun_out/s2o; mkdir -p $O
for pass in 1 2; do for b in base b2 b1 b1st6; do timeout 120 scripts/sb_$b.bin; done; done > $O/sb.txt 2>&1; cat $O/sb.txt
