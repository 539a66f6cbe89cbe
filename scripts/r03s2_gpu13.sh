#!/bin/bash
O=gpurun_out/s2m; mkdir -p $O
timeout 120 scripts/wread.bin > $O/wread.txt 2>&1; cat $O/wread.txt
