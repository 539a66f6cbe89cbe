#!/bin/bash
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s5; mkdir -p $O
{ for v in x_mul_scalar x_mul_twice x_fma_scalar x_mul_nop; do MXI_VERBOSE=1 MXI_SHOW=1 MXI_HIST=6 ./mxi $v.hsaco 64 60; done; } > $O/mxi_variants.txt 2>&1
cat $O/mxi_variants.txt
