#!/bin/bash
# round 2, GPU call 3: why the streamed contraction sits at 55 us -- grid scaling, operand layouts, SQ / TCC counters
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python scripts/gemm_diag.py 2>&1 | tail -14) > $O/gemm_diag.txt
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1
T=$GRAFT_REPO_ROOT/scripts/gemm_pmc_target.py
for lay in plain tiled; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES -d /tmp/pmc_sq1_$lay -o run -- python $T $lay > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT -d /tmp/pmc_sq2_$lay -o run -- python $T $lay > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d /tmp/pmc_tcc_$lay -o run -- python $T $lay > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f_$lay -o run -- python $T $lay > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum -d /tmp/pmc_tcp_$lay -o run -- python $T $lay > /dev/null 2>&1
  (cd $GRAFT_REPO_ROOT && python scripts/pmc_summary.py /tmp/pmc_sq1_$lay /tmp/pmc_sq2_$lay /tmp/pmc_tcc_$lay /tmp/pmc_f_$lay /tmp/pmc_tcp_$lay 2>&1 | grep -A40 "k_modegemm_dma") > $GRAFT_REPO_ROOT/$O/pmc_$lay.txt
done
cd $GRAFT_REPO_ROOT
cat $O/gemm_diag.txt; cat $O/pmc_plain.txt; cat $O/pmc_tiled.txt; grep -c . $O/counters_list.txt
