#!/bin/bash
# round 5, GPU session 1: full GPU tier + A-B of the scratch-removal changes
O=gpurun_out/r05_s1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
python scripts/f2p_time.py > $O/f2p_time.txt 2>&1; tail -20 $O/f2p_time.txt
ODD_CASES=2,4,6 ODD_NO_REF=1 python scripts/odd_sizes_time.py > $O/odd_occ2.txt 2>&1; cat $O/odd_occ2.txt
SC_ENGINE_LIB=$PWD/neuraloperator_amd/libsc_engine_occ3.so ODD_CASES=2,4,6 ODD_NO_REF=1 python scripts/odd_sizes_time.py > $O/odd_occ3.txt 2>&1; cat $O/odd_occ3.txt
