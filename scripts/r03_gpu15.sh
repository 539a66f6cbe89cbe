#!/bin/bash
# round 3, GPU call 15: the ragged 36 x 36 per-mode products of the Tucker chain on the streamed matrix-core kernel
# (SC_G8_FILL4=2: tiles filled >= 1/2 qualify), GaLore replay on the engine, one-rank mode-parallel variants
O=gpurun_out/r3n; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "galore or variants_on_device" 2>&1 | tail -4) > $O/pytest_new.log
cat $O/pytest_new.log
(SC_G8_FILL4=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -x -q -k "tucker or tfno or factor" 2>&1 | tail -4) > $O/pytest_tucker_fill2.log
cat $O/pytest_tucker_fill2.log
for f in 3 2 3 2; do
  echo "SC_G8_FILL4=$f" >> $O/tfno_time.txt
  (SC_G8_FILL4=$f timeout 200 python scripts/tfno_time.py factorized 2>&1 | tail -1) >> $O/tfno_time.txt
done
cat $O/tfno_time.txt
cd /tmp && export TMPDIR=/tmp
SC_G8_FILL4=2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/tfno_kernel_stats_fill2.txt 2>&1
head -24 $O/tfno_kernel_stats_fill2.txt | cut -c1-170
