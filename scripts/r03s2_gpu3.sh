#!/bin/bash
# round 3, session 2, GPU call 3: persistent forward (last image's prefetch -> shared hot rows) and persistent inverse
O=gpurun_out/s2c; mkdir -p $O
for pass in 1 2 3; do
  for b in r3base persist2 persist2_self; do timeout 60 scripts/f3ab_$b.bin 200; done
  F3_INV_GRID=768 timeout 60 scripts/f3ab_persist2.bin 200
  F3_INV_GRID=512 timeout 60 scripts/f3ab_persist2.bin 200
  F3_INV_GRID=2048 timeout 60 scripts/f3ab_persist2.bin 200
done > $O/f3ab.txt 2>&1
cat $O/f3ab.txt
