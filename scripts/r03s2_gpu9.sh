#!/bin/bash
O=gpurun_out/s2i; mkdir -p $O
timeout 300 python scripts/host_profile.py > $O/host_dense.txt 2>&1; grep -v "^$" $O/host_dense.txt | head -75 | cut -c1-160
timeout 300 python scripts/host_profile.py tucker > $O/host_tucker.txt 2>&1; grep "issue" $O/host_tucker.txt
