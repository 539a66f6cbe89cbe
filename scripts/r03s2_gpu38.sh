#!/bin/bash
O=gpurun_out/s2ar; mkdir -p $O
export TFNO_WARM=300 TFNO_STEPS=300
for i in 1 2 3 4; do
  for v in side serial; do
    if [ $v = serial ]; then export SC_NO_SIDE_STREAM=1; else unset SC_NO_SIDE_STREAM; fi
    echo -n "$v: "; timeout 300 python scripts/tfno_time.py factorized 2>&1 | grep -v amdgpu | tail -1
  done
done 2>&1 | tee $O/tfno_two_streams_ab.txt
