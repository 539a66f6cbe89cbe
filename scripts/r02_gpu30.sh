#!/bin/bash
# round 2, GPU call 30: small batches on the streamed contraction (rows of one 32-row tile clamped): FNO3d 128^3 (B = 8)
# and 1024^2 / hidden 128 (B = 4) against the VALU kernel they take today
O=gpurun_out/r2ad; mkdir -p $O
P=neuraloperator_amd
SHAPE=8,32,128,128,128,32,32,17 KINDS=fwd,seq,pair,step ROUNDS=5 REPS=20 timeout 200 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_smallp8.so > $O/smallp_3d.txt 2> $O/smallp_3d.err
cat $O/smallp_3d.txt
SHAPE=4,128,1024,1024,256,129 KINDS=fwd,seq,pair,step ROUNDS=3 REPS=6 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_smallp4.so > $O/smallp_1024.txt 2> $O/smallp_1024.err
cat $O/smallp_1024.txt; tail -2 $O/smallp_1024.err
