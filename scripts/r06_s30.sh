#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s30; mkdir -p $O; D=scripts/session/plinx
{ for v in $D/*.hsaco; do $D/plinx_bench $v; done; } 2>&1 | grep -v amdgpu.ids > $O/plinx_variants.txt
cat $O/plinx_variants.txt
