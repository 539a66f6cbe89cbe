#!/bin/bash
# round 3, session 2: the Tucker chain's backward launches on two streams -- parity (Tucker / graph tests), step time A-B
O=gpurun_out/s2ap; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_graph.py tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "graph or tucker or Tucker or tfno or factor or reproducible" 2>&1 | tail -3 | tee $O/pytest.txt
for i in 1 2 3; do
  for v in side serial; do
    if [ $v = serial ]; then export SC_NO_SIDE_STREAM=1; else unset SC_NO_SIDE_STREAM; fi
    echo -n "$v: "; timeout 300 python scripts/tfno_time.py factorized 2>&1 | grep -v amdgpu | tail -1
  done
done 2>&1 | tee $O/tfno_two_streams_ab.txt
