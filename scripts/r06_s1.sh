#!/bin/bash
# round 6, GPU call 1: the H = 64 question -- which build variant of k_fft2d_inv_mx<64> still differs from run to run
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s1; mkdir -p $O
{
MXI_VERBOSE=1 ./mxi base.hsaco 64 40
for g in 256 320 384 512 768; do ./mxi base.hsaco 64 30 $g; done
for v in *.hsaco; do [ $v = base.hsaco ] || ./mxi $v 64 40; done
./mxi base.hsaco 128 100; ./mxi base.hsaco 256 100
} > $O/mxi_variants.txt 2>&1
tail -60 $O/mxi_variants.txt
