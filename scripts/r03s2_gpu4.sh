#!/bin/bash
# round 3, session 2, GPU call 4: persistent inverse FFT kernel variants (tracked / untracked request, register budget,
# unrolled rounds, workgroups per compute unit)
O=gpurun_out/s2d; mkdir -p $O
for pass in 1 2 3; do
  for b in r3base inv3_trk inv3_trk_occ3 inv3_asm_occ3 inv3_asm_occ3_unr inv3_trk_occ3_unr; do timeout 60 scripts/f3ab_$b.bin 200; done
  for g in 640 704 736 800 832 896; do F3_INV_GRID=$g timeout 60 scripts/f3ab_inv3_asm_occ3.bin 200; done
done > $O/f3ab.txt 2>&1
cat $O/f3ab.txt
