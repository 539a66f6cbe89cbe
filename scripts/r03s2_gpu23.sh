#!/bin/bash
O=gpurun_out/s2w; mkdir -p $O
for pass in 1 2 3; do for b in tg tl1 tl2; do timeout 100 scripts/pl128_$b.bin 30; done; done > $O/pl128_tab2.txt 2>&1; cat $O/pl128_tab2.txt
