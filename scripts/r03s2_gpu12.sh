#!/bin/bash
O=gpurun_out/s2l; mkdir -p $O
timeout 120 scripts/wgrad_store.bin > $O/wgrad_store.txt 2>&1; cat $O/wgrad_store.txt
