#!/bin/bash
cd "$(dirname "$0")/session/pmlp"; O=/root/repo/gpurun_out/r06_s12; mkdir -p $O
./ovl > $O/mfma_valu_overlap.txt 2>&1; cat $O/mfma_valu_overlap.txt
