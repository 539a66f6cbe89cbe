#!/bin/bash
# round 2, GPU call 32: hidden 128 at the metric grid (B = 32): the backward pair launch (narrow shape for both jobs)
# against two launches of the wide shape each job would choose alone (-DSC_G8_PAIR_NARROW_ONLY)
O=gpurun_out/r2af; mkdir -p $O
P=neuraloperator_amd
SHAPE=32,128,256,256,64,33 KINDS=seq,pair,bwd,step ROUNDS=5 REPS=20 timeout 200 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_narrowonly.so > $O/pair_c128.txt 2> $O/pair_c128.err
cat $O/pair_c128.txt
