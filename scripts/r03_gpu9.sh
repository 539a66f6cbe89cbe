#!/bin/bash
# round 3, GPU call 9: branch-free prefetch rings (counted vmcnt waits) in k_modegemm_sb and k_modegemm_bfac:
# configs[4]'s three contractions, the TFNO step
O=gpurun_out/r3i; mkdir -p $O
P=neuraloperator_amd
C5=4,128,1024,1024,256,129
SHAPE=$C5 KINDS=fwd,gx,gw,step ROUNDS=3 REPS=5 timeout 300 python scripts/pair_ab.py $P/libsc_engine.so 2>&1 | grep -v amdgpu.ids | tail -2 > $O/sb_c5.txt; cat $O/sb_c5.txt
timeout 200 python scripts/tfno_time.py factorized > $O/tfno_time.txt 2>&1; tail -1 $O/tfno_time.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tfno -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1)
python scripts/rocprof_summary.py /tmp/prof_tfno > $O/tfno_kernel_stats.txt 2>&1; head -14 $O/tfno_kernel_stats.txt | cut -c1-170
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "small_batch or tucker or tfno or golden or C5" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
