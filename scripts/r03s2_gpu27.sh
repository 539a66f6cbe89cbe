#!/bin/bash
# round 3, session 2, GPU call 27: the multi-rank launch contract on ONE GPU (gloo, SC_BENCH_SHARE_GPU): 2 and 8 ranks
O=gpurun_out/s2ab; mkdir -p $O
for n in 2 8; do
  SC_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 5 --warmup 2 --no-pmc > $O/bench_share$n.json 2> $O/bench_share$n.err
  echo "rc=$?"; head -c 900 $O/bench_share$n.json; echo; tail -2 $O/bench_share$n.err | cut -c1-200
done
