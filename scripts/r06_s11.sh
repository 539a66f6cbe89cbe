#!/bin/bash
cd "$(dirname "$0")/session/pmlp"; O=/root/repo/gpurun_out/r06_s11; mkdir -p $O
{ ./pmlp_bench nw4.hsaco 4 0; ./pmlp_bench nw8.hsaco 8 0; ./pmlp_bench nw4_lin.hsaco 4 1; ./pmlp_bench nw8_lin.hsaco 8 1; ./pmlp_bench nw4.hsaco 4 0 512;  ./pmlp_bench nw4.hsaco 4 0; } > $O/pmlp_variants.txt 2>&1
cat $O/pmlp_variants.txt
