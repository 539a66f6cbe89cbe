#!/bin/bash
# round 3, GPU call 1: (a) A-B of the forward / inverse FFT kernel variants (stand-alone harness, settled clocks),
# (b) nt-store shape ubench, (c) engine-level A-B (transforms + whole step) of the round-2 row stage against the
# register reduce-scatter, (d) the GPU tier incl. the new at-config cases, (e) the default bench line with the new extras
O=gpurun_out/r3a; mkdir -p $O
for pass in 1 2; do
for b in r2base new swaponly ldsonly new_pf2occ3 new_noload r2base_noload new_plain1 new_plain2 new_plain4 new_plainio; do
  timeout 60 scripts/f3ab_$b.bin 200
done; done > $O/f3ab.txt 2>&1
cat $O/f3ab.txt
timeout 120 scripts/ubench_ntw.bin > $O/ntw.txt 2>&1; cat $O/ntw.txt
P=neuraloperator_amd
KINDS=tf,ti,fwd,pair,step ROUNDS=9 REPS=40 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_r2base.so $P/libsc_engine_swaponly.so $P/libsc_engine_ldsonly.so > $O/engine_ab.txt 2> $O/engine_ab.err
cat $O/engine_ab.txt; tail -3 $O/engine_ab.err
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/gpu_tier.txt 2>&1; tail -30 $O/gpu_tier.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json; tail -3 $O/bench_default.err
