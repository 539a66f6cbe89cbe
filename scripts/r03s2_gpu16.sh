#!/bin/bash
O=gpurun_out/s2p; mkdir -p $O
for pass in 1 2; do for b in bwd3 bwd2 bwd4; do timeout 120 scripts/sb_$b.bin; done; done > $O/sb.txt 2>&1; cat $O/sb.txt
