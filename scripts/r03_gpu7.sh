#!/bin/bash
# round 3, GPU call 7: (a) step-level A-B of prefetch depth 2 / three workgroups per CU in the forward FFT kernel,
# (b) the factor-operand contraction kernel (k_modegemm_bfac) in the TFNO step: time, kernel stats, parity
O=gpurun_out/r3g; mkdir -p $O
P=neuraloperator_amd
KINDS=tf,ti,step ROUNDS=21 REPS=40 timeout 400 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_pf2occ3.so 2>&1 | grep -v amdgpu.ids | tail -4 > $O/step_ab.txt
cat $O/step_ab.txt
timeout 200 python scripts/tfno_time.py factorized > $O/tfno_time.txt 2>&1; tail -1 $O/tfno_time.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tfno -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1)
python scripts/rocprof_summary.py /tmp/prof_tfno > $O/tfno_kernel_stats.txt 2>&1; head -16 $O/tfno_kernel_stats.txt | cut -c1-170
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "golden or tucker or tfno or cp_ or factor or spherical or mode_parallel" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
