#!/bin/bash
# round 3, GPU call 12: SQ counters of the TFNO step's kernels (where does k_modegemm_msum / k_tucker_modes_* / the
# register-staged MFMA kernel spend its wave cycles?)
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/tfno_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from neuraloperator_amd import SpectralConv
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(32, 64, 256, 256, device=dev, requires_grad=True)
g = torch.randn(32, 64, 256, 256, device=dev)
conv = SpectralConv(64, 64, (64, 64), factorization="Tucker", rank=0.1, implementation="factorized").to(dev)
for _ in range(3):
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    conv(x).backward(g)
torch.cuda.synchronize()
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d /tmp/pmc_t1 -o run -- python /tmp/tfno_one.py > /dev/null 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pmc_t2 -o run -- python /tmp/tfno_one.py > /dev/null 2>&1)
python scripts/pmc_summary.py /tmp/pmc_t1 /tmp/pmc_t2 > $O/tfno_pmc.txt 2>&1
grep -A17 "k_modegemm_msum\|k_tucker_modes\|k_modegemm_mfma<2\|k_modegemm_bfac<9, false, false" $O/tfno_pmc.txt | head -120
