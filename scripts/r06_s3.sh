#!/bin/bash
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s3; mkdir -p $O
{ for v in e1_swap e2_hi e3_dup; do MXI_VERBOSE=1 MXI_SHOW=2 MXI_HIST=16 ./mxi $v.hsaco 64 20; done; } > $O/mxi_variants.txt 2>&1
cat $O/mxi_variants.txt
