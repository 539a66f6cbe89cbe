#!/bin/bash
# round 3, GPU call 2: (a) the forward FFT kernel's load stream alone and its prefetch-depth / occupancy variants,
# (b) the small-extent streaming contraction (k_modegemm_sb) against the kernels it replaces at BASELINE configs[4]
# (B = 4, hidden 128, 1024^2) and configs[3] (FNO3d 128^3, B = 8), (c) the GPU tests that changed
O=gpurun_out/r3b; mkdir -p $O
timeout 120 scripts/ubench_access.bin > $O/access.txt 2>&1; grep -E "rd_gen3|rd_dword|wr_dword|copy" $O/access.txt
for pass in 1 2; do
for b in r2base new new_pf2occ4 new_pf2occ3 new_pf4occ2 new_pf4occ3; do timeout 60 scripts/f3ab_$b.bin 200; done; done > $O/f3ab.txt 2>&1
cat $O/f3ab.txt
P=neuraloperator_amd
C5=4,128,1024,1024,256,129
C4=8,32,128,128,128,32,32,17
for v in 0 4; do echo "== configs[4] shape, SC_SB_MAX=$v"; SC_SB_MAX=$v SHAPE=$C5 KINDS=fwd,seq,step ROUNDS=3 REPS=5 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so 2>&1 | tail -4; done > $O/sb_c5.txt 2>&1
echo "== configs[4] shape, SC_SB_MAX=4, plain C stores" >> $O/sb_c5.txt
SC_SB_MAX=4 SC_SB_PLAIN_C=1 SHAPE=$C5 KINDS=seq,step ROUNDS=3 REPS=5 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so 2>&1 | tail -4 >> $O/sb_c5.txt
cat $O/sb_c5.txt
for v in 4 8; do echo "== configs[3] shape, SC_SB_MAX=$v"; SC_SB_MAX=$v SHAPE=$C4 KINDS=fwd,seq,pair,step ROUNDS=5 REPS=10 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so 2>&1 | tail -4; done > $O/sb_c4.txt 2>&1
cat $O/sb_c4.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_batch_contractions or mode_parallel_layer_on_device or spatial_parallel_layer or optimizer or fused_block or pointwise or spherical or bf16 or galore or graph" > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
