#!/bin/bash
# round 3, session 2, GPU call 5: persistent 128 x 128 plane kernels (FNO3d 128^3) against one plane per workgroup;
# final persistent inverse 2-D kernel once more
O=gpurun_out/s2e; mkdir -p $O
for pass in 1 2 3; do
  timeout 100 scripts/pl128_r3base.bin 30
  timeout 100 scripts/pl128_persist.bin 30
  for g in 512 1024 1536 2048 32768; do PL_GRID=$g timeout 100 scripts/pl128_persist.bin 30; done
done > $O/pl128.txt 2>&1
cat $O/pl128.txt
for pass in 1 2; do for b in r3base final; do timeout 60 scripts/f3ab_$b.bin 200; done; done > $O/f3ab.txt 2>&1; cat $O/f3ab.txt
