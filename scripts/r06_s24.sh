#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s24; mkdir -p $O; D=scripts/session/pmlp
{ for v in b_base b_nomfma b_nogelu b_nostore b_noload b_nomem b_memonly b_noatomic b_mfmaonly; do $D/pmlp_bench $D/$v.hsaco 4 1 256 10; done; } 2>&1 | grep -v amdgpu.ids > $O/pmlp_bwd_ablate.txt
cat $O/pmlp_bwd_ablate.txt
