#!/bin/bash
# round 3, GPU call 21: counters of the factor-matrix kernels (matrix pipe busy, LDS waits / conflicts)
O=gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d /tmp/pmc_t1 -o run -- python $GRAFT_REPO_ROOT/scripts/fmx_time.py > /dev/null 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA -d /tmp/pmc_t2 -o run -- python $GRAFT_REPO_ROOT/scripts/fmx_time.py > /dev/null 2>&1)
python scripts/pmc_summary.py /tmp/pmc_t1 /tmp/pmc_t2 > $O/fmx_pmc.txt 2>&1
grep -A18 "k_modegemm_bfac_mx<16\|k_modegemm_msum_mx" $O/fmx_pmc.txt | head -80
