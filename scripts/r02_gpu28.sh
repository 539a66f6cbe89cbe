#!/bin/bash
# round 2, GPU call 28: the forward FFT kernel's compile-time options re-measured at SETTLED clocks (round 1 chose
# them from short cold runs): workgroups per CU 4 / 3, row prefetch depth 1 / 2, first-stage twiddle form
O=gpurun_out/r2ab; mkdir -p $O
P=neuraloperator_amd
KINDS=tf,ti,step ROUNDS=9 REPS=40 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_occ3.so $P/libsc_engine_pf2.so $P/libsc_engine_occ4pf2.so $P/libsc_engine_tw1.so > $O/fft_opts_ab.txt 2> $O/fft_opts_ab.err
cat $O/fft_opts_ab.txt
