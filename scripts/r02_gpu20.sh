#!/bin/bash
# round 2, GPU call 20: is the module step host-bound?  enqueue vs total time (module path / raw C-ABI), graph replay
O=gpurun_out/r2t; mkdir -p $O
timeout 200 python scripts/host_overhead.py > $O/host_overhead.txt 2> $O/host_overhead.err
cat $O/host_overhead.txt
timeout 200 python scripts/graph_time.py > $O/graph_time.txt 2> $O/graph_time.err
cat $O/graph_time.txt
