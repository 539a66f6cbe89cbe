#!/bin/bash
# round 2, GPU call 26: ablation of the streamed contraction at settled clocks: without the MFMAs, without the C stores, without both
O=gpurun_out/r2z; mkdir -p $O
P=neuraloperator_amd
ROUNDS=7 REPS=40 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_nomfma.so $P/libsc_engine_nostore.so $P/libsc_engine_noboth.so > $O/ablate.txt 2> $O/ablate.err
cat $O/ablate.txt
