#!/bin/bash
# round 3, session 2, GPU call 2: forward FFT kernel, persistent workgroups with cross-image prefetch -- A-B against the
# one-image-per-workgroup form of the last commit (stand-alone harness, settled clocks, two interleaved passes)
O=gpurun_out/s2b; mkdir -p $O
for pass in 1 2 3; do
  for b in r3base persist persist_o2d4 persist_o4d1 persist_o4d2; do timeout 60 scripts/f3ab_$b.bin 200; done
  F3_FWD_GRID=512 timeout 60 scripts/f3ab_persist.bin 200
  F3_FWD_GRID=1024 timeout 60 scripts/f3ab_persist.bin 200
  F3_FWD_GRID=2048 timeout 60 scripts/f3ab_persist.bin 200
  F3_FWD_GRID=256 timeout 60 scripts/f3ab_persist_o2d4.bin 200
  F3_FWD_GRID=1024 timeout 60 scripts/f3ab_persist_o2d4.bin 200
done > $O/f3ab.txt 2>&1
cat $O/f3ab.txt
