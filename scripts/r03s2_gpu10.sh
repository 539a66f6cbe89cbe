#!/bin/bash
# round 3, session 2, GPU call 10: native Tucker chain -- TFNO parity tests, host issue time, TFNO step time
O=gpurun_out/s2j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_at_config.py tests/test_gpu_parity.py -m gpu -x -q -k "tucker or tfno or factorized" 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 300 python scripts/host_profile.py tucker > $O/host_tucker.txt 2>&1; grep -v "^$" $O/host_tucker.txt | head -40 | cut -c1-150
for i in 1 2; do timeout 200 python scripts/tfno_time.py 2>&1 | tail -2; done | tee $O/tfno_time.txt
