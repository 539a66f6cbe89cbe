#!/bin/bash
# round 6, GPU call 2: the H = 64 question -- which entries lose a part (histogram), and nops around the mask SALU ops
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s2; mkdir -p $O
{
MXI_VERBOSE=1 MXI_SHOW=40 ./mxi base.hsaco 64 40
for v in t*.hsaco; do ./mxi $v 64 60; done
} > $O/mxi_variants.txt 2>&1
tail -70 $O/mxi_variants.txt
