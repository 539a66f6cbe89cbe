#!/bin/bash
# round 3, session 2: deeper LDS-DMA ring for contraction launches that do not fill the chip -- parity + stage times A-B
O=gpurun_out/s2as; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "oracle or golden" 2>&1 | tail -2 | tee $O/pytest.txt
for wl in fno2d_64_m32_c64_b64 fno2d_128_m32_c64_b32 fno3d_64_m16_c32_b8; do
  for v in 1 0 1 0; do
    SC_G8_DEEP=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --workload $wl --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.load(open('$O/b.json')); print('$wl deep=$v', d['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k:v['ms'] for k,v in d['stages'].items() if 'contract' in k})"
  done
done 2>&1 | tee $O/deep_ring_ab.txt
