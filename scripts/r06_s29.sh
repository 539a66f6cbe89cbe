#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s29; mkdir -p $O; D=scripts/session/pmlp
{ $D/pmlp_bench $D/b_base.hsaco 4 1 256 10; $D/pmlp_bench $D/c_nw4_a2m.hsaco 4 1 256 10; $D/pmlp_bench $D/c_nw8.hsaco 8 1 256 10; $D/pmlp_bench $D/c_nw8_a2m.hsaco 8 1 256 10; } 2>&1 | grep -v amdgpu.ids > $O/pmlp_nw8.txt
cat $O/pmlp_nw8.txt
