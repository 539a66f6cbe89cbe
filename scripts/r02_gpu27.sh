#!/bin/bash
# round 2, GPU call 27: three real products per complex product on the narrow streamed contraction (default) against
# the four-product build (-DSC_G8_NO_3M); the contraction parity tests at the metric shape
O=gpurun_out/r2aa; mkdir -p $O
P=neuraloperator_amd
ROUNDS=9 REPS=40 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_no3m.so > $O/m3_ab.txt 2> $O/m3_ab.err
cat $O/m3_ab.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "contraction or gemm or at_config" -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" > $O/pytest_gemm.log
cat $O/pytest_gemm.log
