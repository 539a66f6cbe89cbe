#!/bin/bash
# round 2, GPU call 23: step time against the number of steps since start-up / since an idle period (clock ramp)
O=gpurun_out/r2w; mkdir -p $O
timeout 300 python scripts/clock_ramp.py > $O/clock_ramp.txt 2> $O/clock_ramp.err
cat $O/clock_ramp.txt
