#!/bin/bash
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s6; mkdir -p $O
{ for v in y_commute y_fma_swap; do MXI_VERBOSE=1 MXI_SHOW=1 MXI_HIST=6 ./mxi $v.hsaco 64 100; done; ./mxi x_mul_scalar.hsaco 64 2000; } > $O/mxi_variants.txt 2>&1
cat $O/mxi_variants.txt
