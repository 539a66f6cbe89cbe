#!/bin/bash
# round 2, GPU call 19: A-B of the backward pair launch against the launch sequence and the tiles-per-workgroup variants
# (call 18 kept only the tail of the script's stderr)
O=gpurun_out/r2s; mkdir -p $O
P=neuraloperator_amd
ROUNDS=9 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_nopair.so $P/libsc_engine_bpw1.so $P/libsc_engine_bpw2.so > $O/pair_ab.txt 2> $O/pair_ab.err
cat $O/pair_ab.txt
