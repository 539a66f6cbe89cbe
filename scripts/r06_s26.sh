#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s26; mkdir -p $O
scripts/session/x6/x6 2>&1 | grep -v amdgpu.ids > $O/bf16x6.txt; cat $O/bf16x6.txt
