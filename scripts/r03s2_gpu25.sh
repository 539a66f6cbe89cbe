#!/bin/bash
O=gpurun_out/s2y; mkdir -p $O
for pass in 1 2 3 4; do for b in final tabs; do timeout 60 scripts/f3ab_$b.bin 200; done; done > $O/f3ab_tabs.txt 2>&1; cat $O/f3ab_tabs.txt
