#!/bin/bash
# round 3, session 2: the pointwise block pass (k_pblock_fwd) -- parity of the fused block, block step time A-B, grid sweep
O=gpurun_out/s2bg; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "block or pointwise" 2>&1 | tail -2 | tee $O/pytest.txt
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export SC_BLOCK_NO_PBLOCK=1; else unset SC_BLOCK_NO_PBLOCK; fi
    echo -n "$v: "; python scripts/block_step_profile.py 2>&1 | tail -1
  done
done 2>&1 | tee $O/block_ab.txt
unset SC_BLOCK_NO_PBLOCK
for n in 512 768 1024 2048; do echo -n "pblock wgs=$n: "; SC_PBLOCK_FWD_WGS=$n python scripts/block_step_profile.py 2>&1 | tail -1; done 2>&1 | tee -a $O/block_ab.txt
