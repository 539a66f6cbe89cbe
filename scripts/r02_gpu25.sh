#!/bin/bash
# round 2, GPU call 25: A-B of the narrow contraction shape: <SUB 2, D 3> (48 KiB, 3 workgroups per CU) against
# <SUB 1, D 5> (40 KiB, 4 per CU) and <SUB 1, D 4> (32 KiB, 5 per CU)
O=gpurun_out/r2y; mkdir -p $O
P=neuraloperator_amd
ROUNDS=9 REPS=40 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_shape1.so $P/libsc_engine_shape2.so > $O/shape_ab.txt 2> $O/shape_ab.err
cat $O/shape_ab.txt
