#!/bin/bash
# round 3, GPU call 3: (a) wave arrangements of k_modegemm_sb at BASELINE configs[4]'s contraction shapes,
# (b) the mode-parallel layer with 8 ranks on ONE GPU over gloo (launch contract + `collectives` of the bench line;
# RCCL itself needs the 8-GPU tier), (c) the whole GPU tier
O=gpurun_out/r3c; mkdir -p $O
P=neuraloperator_amd
C5=4,128,1024,1024,256,129
for wm in 1 4; do echo "== configs[4] shape, SC_SB_WM=$wm (1: waves over column tiles of one 128-mode tile; 4: over 512 contiguous modes)"; SC_SB_WM=$wm SHAPE=$C5 KINDS=fwd,gx,gw,step ROUNDS=3 REPS=5 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so 2>&1 | tail -2; done > $O/sb_arrangement.txt 2>&1
cat $O/sb_arrangement.txt
export SC_BENCH_SHARE_GPU=1
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 3 --warmup 1 --settle-ms 0 --no-extras --stage-iters 2 > $O/share8_metric.json 2> $O/share8_metric.err
cat $O/share8_metric.json | head -c 1500; echo; tail -2 $O/share8_metric.err
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 8 --workload fno3d_128_m32_c32_b8 --parallel modeshard --steps 3 --warmup 1 --settle-ms 0 --no-extras --stage-iters 2 > $O/share8_fno3d.json 2> $O/share8_fno3d.err
cat $O/share8_fno3d.json | head -c 1500; echo; tail -2 $O/share8_fno3d.err
unset SC_BENCH_SHARE_GPU
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/gpu_tier.txt 2>&1; tail -12 $O/gpu_tier.txt
