#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s23; mkdir -p $O; D=scripts/session/pblock
{ for v in base nomfma nogelu nostore noload nomem memonly computeonly_nogelu; do $D/pblock_bench $D/$v.hsaco 512 10 1; done
  for w in 256 768 1024 1280; do $D/pblock_bench $D/base.hsaco $w 10 1; done
  $D/pblock_bench $D/base.hsaco 512 10 0; } 2>&1 | grep -v amdgpu.ids > $O/pblock_ablate.txt
cat $O/pblock_ablate.txt
