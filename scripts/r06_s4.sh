#!/bin/bash
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s4; mkdir -p $O
{ for v in dbg1 dbg2; do MXI_VERBOSE=1 MXI_SHOW=2 MXI_HIST=4 ./mxi $v.hsaco 64 20; done; } > $O/mxi_variants.txt 2>&1
cat $O/mxi_variants.txt
