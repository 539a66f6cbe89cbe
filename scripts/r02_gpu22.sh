#!/bin/bash
# round 2, GPU call 22: raw C-ABI step vs module step, ingredient by ingredient
O=gpurun_out/r2v; mkdir -p $O
timeout 300 python scripts/step_variants.py > $O/step_variants.txt 2> $O/step_variants.err
cat $O/step_variants.txt; tail -3 $O/step_variants.err
