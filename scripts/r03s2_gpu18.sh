#!/bin/bash
O=gpurun_out/s2r; mkdir -p $O
for pass in 1 2; do timeout 120 scripts/sb_conc.bin; done > $O/sb.txt 2>&1; cat $O/sb.txt
