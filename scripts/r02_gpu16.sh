#!/bin/bash
# round 2, GPU call 16: pointwise MLP pass + fused block (tests, timing, kernel stats); mode-summed GEMM after the unroll
O=gpurun_out/r2p; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "pointwise or fused_block or fourier or adamw or tucker or golden" 2>&1 | grep -E "passed|failed|Error" | tail -5) > $O/pytest.log
cat $O/pytest.log
(timeout 300 python scripts/block_time.py 2>&1 | tail -5) > $O/block_time.txt
cat $O/block_time.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/scripts/block_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py /tmp/prof > $O/block_kernel_stats.txt 2>&1
head -24 $O/block_kernel_stats.txt | cut -c1-170
(timeout 200 python scripts/tfno_time.py 2>&1 | tail -2) > $O/tfno_time.txt
cat $O/tfno_time.txt
